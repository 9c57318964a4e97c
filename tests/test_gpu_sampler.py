"""On-device triplet sampler: the distributional contract of the reference's generator
(SURVEY.md Appendix F): each record exactly once per epoch, negatives never positives of the user and
uniform over the rest; counter-based determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _data(seed, NU, NI, NR):
    rng = np.random.default_rng(seed)
    raw = np.zeros(NR, dtype=[("user_id", np.int32), ("item_id", np.int32)])
    raw["user_id"] = rng.integers(0, NU, NR); raw["item_id"] = rng.integers(0, NI, NR)
    return raw


def test_sampler_contract():
    import torch
    from openrec_amd import runtime as rt
    NU, NI, NR = 500, 300, 7001
    raw = _data(0, NU, NI, NR)
    raw[:60]["user_id"] = 3; raw[:60]["item_id"] = np.arange(60)        # a user with 20 % of the items positive
    sm = rt.DeviceSampler(raw, NU, NI)
    dev = torch.device("cuda", 0)
    n = 3 * NR
    u, p, ng = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    sm.pairwise(7, 0, n, u, p, ng); sm.ctx.synchronize()
    u, p, ng = u.cpu().numpy(), p.cpu().numpy(), ng.cpu().numpy()
    rec_key = np.sort(raw["user_id"].astype(np.int64) * NI + raw["item_id"])
    for e in range(3):                                                     # every epoch is a permutation of the records
        sl = slice(e * NR, (e + 1) * NR)
        assert np.array_equal(np.sort(u[sl].astype(np.int64) * NI + p[sl]), rec_key)
    assert not np.array_equal(u[:NR], u[NR:2 * NR])                        # epochs are shuffled differently
    pos = set((raw["user_id"].astype(np.int64) * NI + raw["item_id"]).tolist())
    assert all((int(a) * NI + int(b)) not in pos for a, b in zip(u, ng))   # negatives are never positives
    assert ng.min() >= 0 and ng.max() < NI
    cnt = np.bincount(ng, minlength=NI).astype(np.float64)
    assert abs(cnt.mean() - n / NI) < 1e-9 and cnt.std() < 4 * np.sqrt(n / NI)     # roughly uniform
    # counter-based: any window of the stream can be regenerated, seeds differ
    u2, p2, n2 = (torch.empty(1000, dtype=torch.int32, device=dev) for _ in range(3))
    sm.pairwise(7, 5000, 1000, u2, p2, n2); sm.ctx.synchronize()
    assert np.array_equal(u2.cpu().numpy(), u[5000:6000]) and np.array_equal(n2.cpu().numpy(), ng[5000:6000])
    sm.pairwise(8, 5000, 1000, u2, p2, n2); sm.ctx.synchronize()
    assert not np.array_equal(n2.cpu().numpy(), ng[5000:6000])


def test_sampler_feeds_the_fused_step():
    import torch
    from openrec_amd import runtime as rt
    NU, NI, NR, B, K = 2000, 3000, 50000, 4096, 6
    raw = _data(1, NU, NI, NR)
    ctx = rt.default_context()
    sm = rt.DeviceSampler(raw, NU, NI, ctx)
    dev = torch.device("cuda", 0)
    u, p, n = (torch.empty(K * B, dtype=torch.int32, device=dev) for _ in range(3))
    sm.pairwise(3, 0, K * B, u, p, n)
    U = rt.Table(NU, 64).init_uniform(seed=1); V = rt.Table(NI, 64).init_uniform(seed=2); b = rt.Table(NI, 1).init_uniform(seed=3)
    loss, l2 = rt.pairwise_step("bpr", rt.Optimizer.sgd(0.05), U, V, b, u, p, n, K=K, B=B)   # same stream: no sync needed
    assert np.isfinite(loss).all() and abs(loss[0] - np.log(2)) < 0.01


def test_device_sampler_follows_the_reference_generators_distribution():
    """The oracle of SURVEY.md 8(f) row 1: tests/golden/sampler_hist.npz was minted by running the reference's own
    `_pairwise_generator` (data/dataset.py:7-16, utils.py:82-87, 102-116) for 300 epochs.  The device sampler is
    counter-based, so the SEQUENCE differs by construction; held here: the positives' counts exactly, and the per-user
    histograms of the negatives by chi-square -- two-sample against the reference's and one-sample against the uniform law
    over the user's non-positive items (2187 cells; |z| < 4: a 2 % tilt of the cell probabilities gives z ~ 10)."""
    import torch
    import sampler_stats as ss
    from openrec_amd import runtime as rt
    g, raw, NU, NI, E = ss.load()
    sm = rt.DeviceSampler(raw, NU, NI)
    dev = torch.device("cuda", 0)
    n = E * len(raw)
    u, p, ng = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(3))
    sm.pairwise(2024, 0, n, u, p, ng); sm.ctx.synchronize()
    pos, neg = ss.histograms(u.cpu().numpy(), p.cpu().numpy(), ng.cpu().numpy(), NU, NI)
    z_two, z_one, df = ss.check_against_golden(pos, neg, g, raw, NU, NI, E)
    assert abs(z_two) < 4.0 and abs(z_one) < 4.0, (z_two, z_one, df)
    # a different stream is a different sample of the same law
    sm.pairwise(7, 0, n, u, p, ng); sm.ctx.synchronize()
    pos2, neg2 = ss.histograms(u.cpu().numpy(), p.cpu().numpy(), ng.cpu().numpy(), NU, NI)
    assert not np.array_equal(neg2, neg)
    z_two, z_one, _ = ss.check_against_golden(pos2, neg2, g, raw, NU, NI, E)
    assert abs(z_two) < 4.0 and abs(z_one) < 4.0, (z_two, z_one)


def test_device_pointwise_samplers_follow_the_reference_generators():
    """dataset.py:18-58 on the device (the producers of GMF / WRMF): coin ratio, shuffle-and-pop positives, negatives
    uniform over the non-positive pairs (stratified); groups of a record and distinct other items (per-positive) --
    against the fixture minted from the reference's generators, and fed to the fused pointwise step."""
    import torch
    import sampler_stats as ss
    from openrec_amd import runtime as rt
    g, raw, NU, NI, _ = ss.load()
    sm = rt.DeviceSampler(raw, NU, NI)
    dev = torch.device("cuda", 0)
    n, ratio = int(g["strat_n"]), float(g["strat_ratio"])
    u = torch.empty(n, dtype=torch.int32, device=dev); i = torch.empty_like(u); lab = torch.empty(n, dtype=torch.float32, device=dev)
    # three calls: the stream continues where the previous call stopped
    cuts = [0, 1000, 25000, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        sm.stratified_pointwise(11, lo, hi - lo, ratio, u[lo:hi], i[lo:hi], lab[lo:hi])
    sm.ctx.synchronize()
    r = ss.check_stratified(u.cpu().numpy(), i.cpu().numpy(), lab.cpu().numpy(), g, raw, NU, NI)
    assert all(abs(r[k]) < 4.0 for k in ("got_z_ratio", "got_z_uniform", "z_two")), {k: v for k, v in r.items() if "z" in k}
    u2 = torch.empty_like(u); i2 = torch.empty_like(i); l2 = torch.empty_like(lab)
    sm.stratified_pointwise(11, 0, n, ratio, u2, i2, l2); sm.ctx.synchronize()          # one call = the same stream
    assert torch.equal(u, u2) and torch.equal(i, i2) and torch.equal(lab, l2)
    with pytest.raises(Exception):
        sm.stratified_pointwise(11, 5, 10, ratio, u2, i2, l2)                            # not where the stream stands
    E2, pr = int(g["perpos_epochs"]), float(g["perpos_ratio"])
    m = E2 * len(raw) * (1 + int((1 - pr) / pr))
    u = torch.empty(m, dtype=torch.int32, device=dev); i = torch.empty_like(u); lab = torch.empty(m, dtype=torch.float32, device=dev)
    sm.per_pos_stratified_pointwise(12, 0, m, pr, u, i, lab); sm.ctx.synchronize()
    z = ss.check_per_pos(u.cpu().numpy(), i.cpu().numpy(), lab.cpu().numpy(), g, raw, NU, NI)
    assert all(abs(x) < 4.0 for x in z), z
    w = torch.empty(777, dtype=torch.int32, device=dev); wi = torch.empty_like(w); wl = torch.empty(777, dtype=torch.float32, device=dev)
    sm.per_pos_stratified_pointwise(12, 4321, 777, pr, w, wi, wl); sm.ctx.synchronize()  # counter-based: any window
    assert torch.equal(w, u[4321:4321 + 777]) and torch.equal(wi, i[4321:4321 + 777]) and torch.equal(wl, lab[4321:4321 + 777])
    # ... and straight into the fused pointwise step (same stream, no host copy)
    B, K = 4096, 4
    U = rt.Table(NU, 64).init_uniform(seed=1); V = rt.Table(NI, 64).init_uniform(seed=2); b = rt.Table(NI, 1).init_uniform(seed=3)
    loss, _ = rt.pointwise_step("wrmf", rt.Optimizer.sgd(0.01), U, V, b, None, u[:K * B], i[:K * B], lab[:K * B], K=K, B=B)
    assert np.isfinite(loss).all() and loss[0] > 0


@pytest.mark.parametrize("ratio", [0.05, 1.0 / 3.0, 1.0 / 6.0, 1.0 / 11.0, 0.5, 0.2])
def test_per_pos_group_size_is_the_python_double_quotient(ratio):
    """dataset.py:40: num_negative_per_positive = int((1 - pos_ratio) / pos_ratio) in Python doubles.  In fp32 the quotient
    lands on the other side of an integer for common ratios (0.05 -> 19 instead of 18, 1/3 -> 1 instead of 2, 1/6 -> 4
    instead of 5, 1/11 -> 9 instead of 10): the device sampler must emit the reference's group size and label mix."""
    import torch
    from openrec_amd import runtime as rt
    import sampler_stats as ss
    g, raw, NU, NI, _ = ss.load()
    sm = rt.DeviceSampler(raw, NU, NI)
    nneg = int((1 - ratio) / ratio)
    assert nneg + 1 <= NI
    grp = 1 + nneg
    n = 50 * grp
    dev = torch.device("cuda", 0)
    u = torch.empty(n, dtype=torch.int32, device=dev); i = torch.empty_like(u); lab = torch.empty(n, dtype=torch.float32, device=dev)
    sm.per_pos_stratified_pointwise(3, 0, n, ratio, u, i, lab); sm.ctx.synchronize()
    L = lab.cpu().numpy().reshape(-1, grp)
    assert (L[:, 0] == 1.0).all() and (L[:, 1:] == 0.0).all()
    U = u.cpu().numpy().reshape(-1, grp)
    assert (U == U[:, :1]).all()
