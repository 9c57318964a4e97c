"""Stress of the in-launch duplicate apply (apply blocks + ready-flag hand-off inside the fused
launch): small tables make almost every row duplicated AND urgent in every step; many steps and
several calls are compared step by step with the fp64 oracle."""
import numpy as np
import pytest

from conftest import TOL_ADAM

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", [(3000, 3000, 8192, 24, 64, "sgd"), (2000, 2500, 4096, 20, 128, "adagrad"),
                                 (40000, 40000, 65536, 8, 32, "sgd")])
def test_handoff_under_heavy_duplication(cfg):
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B, K, D, optk = cfg
    rng = np.random.default_rng(B)
    U32 = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V32 = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b32 = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    U, V, b = U32.astype(np.float64), V32.astype(np.float64), b32.astype(np.float64)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    lr = 0.02 / (B / min(NU, NI))                      # lr * multiplicity < 1: rounding is not amplified
    tU = rt.Table(NU, D).write(U32); tV = rt.Table(NI, D).write(V32); tb = rt.Table(NI, 1).write(b32)
    opt = rt.Optimizer.sgd(lr) if optk == "sgd" else rt.Optimizer.adagrad(lr, 0.1, 1e-7)
    oo = orc.SGD(lr) if optk == "sgd" else orc.Adagrad(lr, 0.1, 1e-7)
    for rep in range(2):                               # the call boundary is part of the protocol
        loss, l2 = rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
        for s in range(K):
            ref, _ = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
            assert abs(loss[s] - ref) <= 2e-5 * abs(ref), (rep, s)
        for dev, host in ((tU, U), (tV, V), (tb, b)):
            assert np.abs(dev.read() - host).max() <= 5e-5 * np.abs(host).max(), rep


@pytest.mark.parametrize("fallback", ["1", "2", "8"])
def test_fallback_paths_give_the_same_result(fallback, monkeypatch):
    """The paths taken by tables of >= 2^28 rows (no role bits: atomics + dup_apply launches), without the
    in-launch apply (2) and without the staging plan (8: rows referenced >= 3 times use atomics) are forced
    through ORX_FORCE_FALLBACK and checked like the main path."""
    monkeypatch.setenv("ORX_FORCE_FALLBACK", fallback)
    test_handoff_under_heavy_duplication((3000, 3000, 8192, 12, 64, "sgd"))
    test_handoff_under_heavy_duplication((2000, 2500, 4096, 10, 128, "adagrad"))


def test_batch_larger_than_the_grid_cap():
    """B = 1.2 M triplets: the fused grid is capped at 65536 blocks, so lane groups grid-stride."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B, D = 400000, 500000, 1_200_003, 16
    rng = np.random.default_rng(0)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (2, B)).astype(np.int32); pid = rng.integers(0, NI, (2, B)).astype(np.int32); nid = rng.integers(0, NI, (2, B)).astype(np.int32)
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    loss, l2 = rt.pairwise_step("bpr", rt.Optimizer.sgd(0.05), tU, tV, tb, uid, pid, nid, K=2, B=B)
    oo = orc.SGD(0.05)
    for s in range(2):
        ref, l2r = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
        assert abs(loss[s] - ref) <= 1e-5 * abs(ref) and abs(l2[s] - l2r) <= 1e-5 * abs(l2r)
    assert np.abs(tU.read() - U).max() <= 1e-5 * np.abs(U).max() and np.abs(tV.read() - V).max() <= 1e-5 * np.abs(V).max()


@pytest.mark.parametrize("model,D,optname,K", [("bpr", 64, "sgd", 5), ("ucml", 128, "sgd", 3), ("bpr", 16, "adagrad", 4),
                                               ("bpr", 64, "sgd", 1), ("bpr", 64, "adam", 6), ("ucml", 32, "adam", 4)])
def test_skewed_items_use_staging_and_hot_reduce(model, D, optname, K):
    """Items ~ Zipf(1.05): the hottest row takes ~10 % of the 2B item references of a step (hundreds of
    references -> long staging segments, hot_reduce_kernel), a long tail of rows takes 3..64 (segments summed
    by the apply).  The oracle runs in float64 here: a row that sums ~1000 gradients has no unique fp32
    answer (sequential fp32 summation alone is off by ~1e-4 of the largest term), so both the fp32 oracle
    and the kernels are only comparable to the exact sum."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B = 20000, 6000, 8192
    rng = np.random.default_rng(11)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(NI)                                   # hot rows scattered over the table
    draw = lambda: perm[np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1)].astype(np.int32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = draw(); nid = draw()
    uid[:, :300] = 7                                             # one hot user as well (a long user segment)
    assert np.bincount(pid[0], minlength=NI).max() > 300
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    U, V, b = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
    lr = 0.002
    opt = {"sgd": lambda: rt.Optimizer.sgd(lr), "adagrad": lambda: rt.Optimizer.adagrad(lr), "adam": lambda: rt.Optimizer.adam(lr)}[optname]()
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr), "adam": lambda: orc.AdamTFSparse(lr)}[optname]()
    loss, l2 = rt.pairwise_step(model, opt, tU, tV, tb, uid, pid, nid, K=K, B=B)     # (adam: the lazy rule on staged / tree-reduced rows)
    step = orc.bpr_step if model == "bpr" else (lambda *a: orc.ucml_step(*a, margin=0.5, do_censor=False))
    for s in range(K):
        ref, l2r = step(U, V, b, uid[s], pid[s], nid[s], oo)
        assert abs(loss[s] - ref) <= 2e-5 * abs(ref) and abs(l2[s] - l2r) <= 2e-5 * abs(l2r), (s, loss[s], ref)
    for got, want in ((tU.read(), U), (tV.read(), V), (tb.read(), b)):
        e = np.abs(got - want) / np.abs(want).max()
        if optname != "adam":
            assert e.max() <= 2e-5          # (docstring: ~1000-term fp32 sums against the exact sum)
        else:
            # conftest.TOL_ADAM prices the rounding of ONE 0.05-sized gradient term (3e-9) through lr_1 (1 - beta_1) delta / eps.
            # A hot row's element sums hundreds of terms (delta ~ sqrt(300) * 3e-9 = 5e-8, and the slot order of its
            # references -- hence the fp32 sum -- differs from run to run): where such a sum nearly cancels the bound is
            # 17 x larger, capped by the update itself (<= lr_t * 3.2 = 2e-3 absolute).  All but 1e-4 of the elements sit
            # inside TOL_ADAM, none beyond 17 x.
            assert (e > TOL_ADAM).mean() <= 1e-4 and e.max() <= 17 * TOL_ADAM, (float((e > TOL_ADAM).mean()), float(e.max()))


def test_skewed_items_with_fused_censor():
    """hot rows (staging plan, reduction tree) together with censor_vec fused into the apply"""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B, D, K = 5000, 3000, 4096, 64, 4
    rng = np.random.default_rng(12)
    U = (rng.uniform(-.05, .05, (NU, D)) * 25).astype(np.float32); V = (rng.uniform(-.05, .05, (NI, D)) * 25).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    perm = rng.permutation(NI)
    draw = lambda: perm[np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1)].astype(np.int32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = draw(); nid = draw()
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    U, V, b = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
    loss, l2 = rt.pairwise_step("ucml", rt.Optimizer.sgd(0.001), tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
    oo = orc.SGD(0.001)
    for s in range(K):
        ref, l2r = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
        assert abs(loss[s] - ref) <= 2e-5 * abs(ref) and abs(l2[s] - l2r) <= 2e-5 * abs(l2r), (s, loss[s], ref)
    for got, want in ((tU.read(), U), (tV.read(), V), (tb.read(), b)):
        assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


@pytest.mark.parametrize("cfg", [(3000, 3000, 8192, 12, 64, 0.999), (60000, 50000, 4096, 20, 32, 0.999), (200000, 150000, 16384, 10, 128, 0.999),
                                 (60000, 50000, 4096, 20, 64, 0.95), (3000, 2500, 4096, 12, 128, "ucml+censor"),
                                 (40000, 30000, 8192, 10, 32, "ucml+censor")])
def test_lazy_adam_is_the_dense_decay_adam(cfg, monkeypatch):
    """TF-2.0 Adam moves EVERY row every step (m, v decay; var -= lr_t*m/(sqrt(v)+eps)).  The K-step path applies
    that lazily (a row replays its gradient-free steps when next touched or read); it is compared with the
    fp64 statement of the dense rule step by step, across a call boundary, a mid-run read (flush) and a
    change of learning rate, and with the whole-table-sweep form of the same library (ORX_ADAM_DENSE=1)."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B, K, D, beta2 = cfg                       # beta_2 = 0.95: the replay takes v_rcp instead of Newton steps
    ucml = beta2 == "ucml+censor"                      # UCML with censor_vec after every step (fused into the lazy Adam write-back)
    beta2 = 0.999 if ucml else beta2
    lr0 = 0.002 if beta2 == 0.999 else 0.0005          # (short v memory: larger normalised steps amplify fp32 rounding)
    rng = np.random.default_rng(D + K)
    U32 = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V32 = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b32 = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (3 * K, B)).astype(np.int32); pid = rng.integers(0, NI, (3 * K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (3 * K, B)).astype(np.int32)
    results = {}
    for form in ("lazy", "dense"):
        if form == "dense":
            monkeypatch.setenv("ORX_ADAM_DENSE", "1")
        tU = rt.Table(NU, D).write(U32); tV = rt.Table(NI, D).write(V32); tb = rt.Table(NI, 1).write(b32)
        opt = rt.Optimizer.adam(lr0, 0.9, beta2, 1e-7)
        losses = []
        for rep in range(3):
            sl = slice(rep * K, (rep + 1) * K)
            loss, _ = rt.pairwise_step("ucml" if ucml else "bpr", opt, tU, tV, tb, uid[sl], pid[sl], nid[sl], K=K, B=B, censor=ucml)
            losses.append(loss.copy())
            if rep == 0:
                mid = tV.read().copy()                  # observes the table: every row must be current here
            if rep == 1:
                opt.set_lr(lr0 / 2)
        results[form] = (np.concatenate(losses), mid, tU.read(), tV.read(), tb.read(), opt.slot(tV, 0), opt.slot(tV, 1), opt.slot(tb, 0))
    U, V, b = U32.astype(np.float64), V32.astype(np.float64), b32.astype(np.float64)
    oo = orc.AdamTFSparse(lr0, 0.9, beta2, 1e-7)
    ref_loss = []
    for s in range(3 * K):
        if s == 2 * K:
            oo.lr = lr0 / 2
        if ucml:
            ref, _ = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
        else:
            ref, _ = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
        ref_loss.append(ref)
        if s == K - 1:
            ref_mid = V.copy()
    # (UCML: a handful of rows sit on the hinge / on the censor threshold, where fp32 and the fp64 oracle part by
    #  7e-5 in every form -- sweep, lazy, lazy with separate censor passes alike)
    wtol = 2e-4 if ucml else 5e-5
    for form, (loss, mid, dU, dV, db, mV, vV, mb) in results.items():
        assert np.abs(loss - np.array(ref_loss)).max() <= 2e-5 * np.abs(ref_loss).max(), form
        assert np.abs(mid - ref_mid).max() <= wtol * np.abs(ref_mid).max(), form
        for dev, host in ((dU, U), (dV, V), (db, b), (mV, oo.m["V"]), (mb, oo.m["b"])):
            assert np.abs(dev - host).max() <= wtol * np.abs(host).max(), form
        assert np.abs(vV - oo.v["V"]).max() <= 5e-4 * np.abs(oo.v["V"]).max(), form


def test_lazy_adam_long_gaps_in_the_fused_step(monkeypatch):
    """Tables large relative to the batch: rows wait hundreds of steps between references.  The fused step then runs the
    LONGGAP variant (rows behind by more than 256 steps take a bounded replay of their own: the loop stops when the
    update can no longer change the weight, m and v finish in closed form).  700 steps of 64 triplets on 40 000-row
    tables against the C oracle's whole-table sweeps, and against the plain variant (ORX_ADAM_NO_LONGGAP=1)."""
    from openrec_amd import runtime as rt
    from oracle import c_oracle
    NU = NI = 40000
    D, B, K = 32, 64, 700
    rng = np.random.default_rng(21)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    uid[5, 0] = uid[600, 0] = 7; pid[9, 1] = pid[650, 1] = 11; nid[3, 2] = nid[699, 2] = 13      # known long waits
    got = {}
    for form in ("longgap", "plain"):
        if form == "plain":
            monkeypatch.setenv("ORX_ADAM_NO_LONGGAP", "1")
        tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
        opt = rt.Optimizer.adam(0.002)
        loss, _ = rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
        got[form] = (loss.copy(), tU.read(), tV.read(), tb.read(), opt.slot(tV, 0), opt.slot(tV, 1))
    Uc, Vc, bc = U.copy(), V.copy(), b.copy()
    cpu = c_oracle.PairwiseCPU("bpr", "adam", Uc, Vc, bc, lr=0.002)
    ref = np.array([cpu.step(uid[s], pid[s], nid[s])[0] for s in range(K)])
    # Against the (fp32) C oracle a few dozen rows with nearly cancelling gradients part by up to 1e-3 in EVERY form (sweep,
    # plain lazy, LONGGAP alike -- Adam normalises the gradient away): all but 1 % of the rows within 5e-5, none beyond
    # 5e-3; the two lazy variants agree with each other far more closely.
    for form, (loss, gU, gV, gb, mV, vV) in got.items():
        assert np.abs(loss - ref).max() <= 5e-5 * np.abs(ref).max(), form
        for dev, host in ((gU, Uc), (gV, Vc), (gb, bc.reshape(-1, 1))):
            e = np.abs(dev - host).max(axis=1) / np.abs(host).max()
            assert (e > 5e-5).mean() <= 1e-2 and e.max() < 5e-3, (form, float(e.max()))
        em = np.abs(mV - cpu.m[1]).max(axis=1) / np.abs(cpu.m[1]).max()
        assert (em > 1e-4).mean() <= 1e-2 and em.max() < 2e-2, (form, float(em.max()))
    for x, y in zip(got["longgap"][1:4], got["plain"][1:4]):
        assert np.abs(x - y).max() <= 1e-4 * np.abs(y).max()      # (v_rcp in the bounded loop, carried reciprocal in the merged one)


@pytest.mark.parametrize("form", ["lazy", "dense"])
def test_shared_adam_optimizer_updates_only_the_tables_of_a_step(form, monkeypatch):
    """ONE Adam optimizer driving several models (Keras: apply_gradients touches only the variables it is handed, the
    step counter `iterations` is shared).  Model A = (UA, VA, bA), model B = (UB, VB, bB), model C = (UA, VC, bC) shares
    A's user table.  The lazily-applied rule must not replay another model's steps as decay steps of this model's rows
    (advisor finding, round 1): both forms against the fp64 oracle with one shared AdamTFSparse."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    from conftest import TOL_ADAM
    if form == "dense":
        monkeypatch.setenv("ORX_ADAM_DENSE", "1")
    D, B = 64, 512
    rng = np.random.default_rng(21)
    sizes = dict(UA=700, VA=900, UB=300, VB=400, VC=500)
    host = {k: rng.uniform(-.05, .05, (n, D)).astype(np.float32) for k, n in sizes.items()}
    host.update(bA=rng.uniform(-.05, .05, (900, 1)).astype(np.float32), bB=rng.uniform(-.05, .05, (400, 1)).astype(np.float32),
                bC=rng.uniform(-.05, .05, (500, 1)).astype(np.float32))
    dev = {k: rt.Table(*v.shape).write(v) for k, v in host.items()}
    ref = {k: v.astype(np.float64) for k, v in host.items()}
    models = dict(A=("UA", "VA", "bA"), B=("UB", "VB", "bB"), C=("UA", "VC", "bC"))
    opt = rt.Optimizer.adam(0.002)
    oo = orc.AdamTFSparse(0.002)
    for name, K in (("A", 3), ("B", 2), ("A", 2), ("C", 3), ("B", 1), ("A", 1), ("C", 1)):
        u, v, b = models[name]
        uid = rng.integers(0, sizes[u], (K, B)).astype(np.int32); pid = rng.integers(0, sizes[v], (K, B)).astype(np.int32)
        nid = rng.integers(0, sizes[v], (K, B)).astype(np.int32)
        loss, _ = rt.pairwise_step("bpr", opt, dev[u], dev[v], dev[b], uid, pid, nid, K=K, B=B)
        for s in range(K):
            lr, _ = orc.bpr_step(ref[u], ref[v], ref[b], uid[s], pid[s], nid[s], oo, keys=(u, v, b))
            assert abs(loss[s] - lr) <= 1e-5 * abs(lr), (name, s)
    assert opt.step == 13
    for k in host:
        assert np.abs(dev[k].read() - ref[k]).max() <= TOL_ADAM * np.abs(ref[k]).max(), k
    for k in ("UA", "VA", "VB", "VC"):
        assert np.abs(opt.slot(dev[k], 0) - oo.m[k]).max() <= 1e-5 * np.abs(oo.m[k]).max(), k
