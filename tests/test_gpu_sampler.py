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
