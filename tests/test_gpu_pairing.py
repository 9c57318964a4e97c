"""GPU: the PAIRING path of the exact pairwise step (kernels_plan.hip: plan_range / plan_pair / plan_pack, the LDS gradient
exchange and pair_tail of fused_kernel) held to the oracle AND to the same call with pairing off -- with the assertion that pairs
were really accepted, so that the test cannot pass with the path idle.  ORX_PAIR_ALWAYS=1 keeps pairing on where its
yield is low; ORX_NO_PAIR=1 turns it off.  Covered: every float4 dim with >= 2 triplets per wavefront, BPR and UCML with the fused
censor (positive / negative pairs: censored twice), odd batch sizes (a last position without a buddy), a call long enough to need
the in-launch apply across many steps, skewed ids, an invalid id, and the call forms with and without the mid-call read-back of the
plan counters (ORX_PLAN_WAIT=1).  Reference semantics: tf2_examples/bpr_citeulike.py:35-38 (duplicates summed, then applied)."""
import os

import numpy as np
import pytest

from conftest import rel_err, TOL

pytestmark = pytest.mark.gpu


def _rt():
    from openrec_amd import runtime as rt
    return rt


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _case(seed, NU, NI, B, D, K, zipf=0.0):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32)
    if zipf:
        w = 1.0 / np.arange(1, NI + 1) ** zipf
        cdf = np.cumsum(w / w.sum())
        draw = lambda: np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1).astype(np.int32)
        pid, nid = draw(), draw()
    else:
        pid = rng.integers(0, NI, (K, B)).astype(np.int32)
        nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    return U, V, b, uid, pid, nid


def _run(model, U, V, b, uid, pid, nid, censor=False, opt="sgd"):
    rt = _rt()
    ctx = rt.Context(0)
    tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
    o = rt.Optimizer.sgd(0.05, ctx=ctx) if opt == "sgd" else rt.Optimizer.adagrad(0.05, 0.1, 1e-7, ctx=ctx)
    K, B = uid.shape
    l, l2 = rt.pairwise_step(model, o, tU, tV, tb, uid.reshape(-1), pid.reshape(-1), nid.reshape(-1), K=K, B=B, margin=0.5, censor=censor)
    out = dict(U=tU.read(), V=tV.read(), b=tb.read(), loss=np.asarray(l, np.float64), l2=np.asarray(l2, np.float64),
               pairs=ctx.stat("pairs"), max_dup=ctx.stat("max_dup"))
    return out


def _oracle(model, U, V, b, uid, pid, nid, censor=False, lr=0.05, opt="sgd", dtype=np.float32):
    from oracle import numpy_oracle as orc
    U, V, b = U.astype(dtype), V.astype(dtype), b.astype(dtype)
    opt = orc.SGD(lr=lr) if opt == "sgd" else orc.Adagrad(lr=lr, initial_accumulator_value=0.1, epsilon=1e-7)
    ls = []
    for k in range(uid.shape[0]):
        if model == "bpr":
            ls.append(orc.bpr_step(U, V, b, uid[k], pid[k], nid[k], opt))
        else:
            ls.append(orc.ucml_step(U, V, b, uid[k], pid[k], nid[k], opt, margin=0.5, do_censor=censor))
    return dict(U=U, V=V, b=b, loss=np.array([x[0] for x in ls], np.float64), l2=np.array([x[1] for x in ls], np.float64))


def _check(got, want, what):
    for k in ("U", "V", "b", "loss", "l2"):
        assert rel_err(got[k], want[k]) < TOL, (what, k, rel_err(got[k], want[k]))


@pytest.mark.parametrize("D", [16, 32, 64, 128])
@pytest.mark.parametrize("model,censor", [("bpr", False), ("ucml", True)])
@pytest.mark.parametrize("B", [4096, 4095])
def test_paired_step_equals_unpaired_step_and_oracle(D, model, censor, B):
    # tables ~ 10 x the batch: most duplicated rows are referenced exactly twice (the pairing case)
    U, V, b, uid, pid, nid = _case(11 + D, 40000, 40000, B, D, K=5)
    with env(ORX_PAIR_ALWAYS=1, ORX_NO_PAIR=None):
        paired = _run(model, U, V, b, uid, pid, nid, censor)
    with env(ORX_NO_PAIR=1):
        plain = _run(model, U, V, b, uid, pid, nid, censor)
    want = _oracle(model, U, V, b, uid, pid, nid, censor)
    assert paired["pairs"] > 5 * 50, "pairing was idle: %d accepted pairs" % paired["pairs"]
    assert plain["pairs"] == 0
    assert paired["max_dup"] < plain["max_dup"]                  # paired rows left the apply's list
    _check(paired, want, "paired vs oracle")
    _check(plain, want, "unpaired vs oracle")
    _check(paired, plain, "paired vs unpaired")


def test_adagrad_keeps_the_deposit_path():
    """pairing is SGD-only: with Adagrad the pair tail's accumulator traffic and registers cost more than the deposits they
    replace (profiles/r5_adagrad_pairing_ab.txt: kernel 51.5 against 49.6 us, step 62.1 against 57.7 on one box)"""
    U, V, b, uid, pid, nid = _case(41, 40000, 40000, 4096, 64, K=3)
    with env(ORX_PAIR_ALWAYS=1, ORX_NO_PAIR=None):
        got = _run("bpr", U, V, b, uid, pid, nid, opt="adagrad")
    assert got["pairs"] == 0
    _check(got, _oracle("bpr", U, V, b, uid, pid, nid, opt="adagrad"), "adagrad vs oracle")


def test_paired_steps_in_a_long_call_and_without_the_read_back():
    """K = 70: the in-launch apply runs across many steps of one call (urgent marks, dead list entries); the second call of a context
    takes the form that does not wait for the plan's counters -- same results as with the read-back, and as the oracle"""
    U, V, b, uid, pid, nid = _case(5, 50000, 60000, 8192, 64, K=70)
    res = {}
    for wait in (None, 1):
        with env(ORX_PAIR_ALWAYS=1, ORX_NO_PAIR=None, ORX_PLAN_WAIT=wait):
            rt = _rt()
            ctx = rt.Context(0)
            tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
            o = rt.Optimizer.sgd(0.05, ctx=ctx)
            l1 = rt.pairwise_step("bpr", o, tU, tV, tb, uid[:35].reshape(-1), pid[:35].reshape(-1), nid[:35].reshape(-1), K=35, B=8192)
            l2 = rt.pairwise_step("bpr", o, tU, tV, tb, uid[35:].reshape(-1), pid[35:].reshape(-1), nid[35:].reshape(-1), K=35, B=8192)
            res[wait] = dict(U=tU.read(), V=tV.read(), b=tb.read(), loss=np.concatenate([l1[0], l2[0]]).astype(np.float64),
                             l2=np.concatenate([l1[1], l2[1]]).astype(np.float64), pairs=ctx.stat("pairs"), nowait=ctx.stat("nowait_calls"))
    assert res[None]["nowait"] == 1 and res[1]["nowait"] == 0        # (the first call of a context always reads back)
    assert res[None]["pairs"] > 35 * 100
    want = _oracle("bpr", U, V, b, uid, pid, nid)
    _check(res[None], want, "no read-back vs oracle")
    _check(res[1], want, "read-back vs oracle")
    _check(res[None], res[1], "no read-back vs read-back")


def test_quiet_call_followed_by_a_skewed_one_stays_exact():
    """a call planned WITHOUT staging (its predecessor was quiet) meets ids with rows referenced hundreds of times: rows referenced
    >= 3 times then take fp32 atomics -- exact up to summation order -- and the call after it plans with the read-back again.
    (lr * references of the hottest row < 1: with SGD's l2 term a row referenced c times moves by lr * c * row per step, and a
    product above 1 amplifies fp32 rounding from step to step whatever computes it)"""
    rt = _rt()
    lr = 0.0005
    U, V, b, uid, pid, nid = _case(9, 30000, 30000, 4096, 64, K=4)
    Uz, Vz, bz, uz, pz, nz = _case(10, 30000, 30000, 4096, 64, K=4, zipf=1.1)
    assert np.bincount(np.concatenate([pz[0], nz[0]])).max() > 300          # really skewed
    ctx = rt.Context(0)
    tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
    o = rt.Optimizer.sgd(lr, ctx=ctx)
    f = lambda a: a.reshape(-1)
    rt.pairwise_step("bpr", o, tU, tV, tb, f(uid), f(pid), f(nid), K=4, B=4096)             # reads back: quiet
    assert ctx.stat("quiet") == 1
    rt.pairwise_step("bpr", o, tU, tV, tb, f(uz), f(pz), f(nz), K=4, B=4096)                # no read-back, skewed ids
    assert ctx.stat("nowait_calls") == 1 and ctx.stat("quiet") == 0
    rt.pairwise_step("bpr", o, tU, tV, tb, f(uz), f(pz), f(nz), K=4, B=4096)                # reads back again (staging on)
    assert ctx.stat("nowait_calls") == 1
    # (against the fp64 oracle: the sum of the ~1000 gradients of the hottest row carries sqrt(1000) ulps in ANY fp32 order, the
    # fp32 oracle's sequential order included)
    want = _oracle("bpr", U, V, b, np.concatenate([uid, uz, uz]), np.concatenate([pid, pz, pz]), np.concatenate([nid, nz, nz]), lr=lr,
                   dtype=np.float64)
    got = dict(U=tU.read(), V=tV.read(), b=tb.read())
    errs = {k: rel_err(got[k], want[k]) for k in ("U", "V", "b")}
    assert all(e < TOL for e in errs.values()), errs


def test_an_invalid_id_never_takes_a_partner_down():
    """a triplet with an out-of-range id is skipped (the reference's gather raises: the call reports ORX_ERR_INDEX); the plan must
    not pair it, or the valid triplet that shares a row with it would lose its own update"""
    rt = _rt()
    NU, NI, B, D = 3000, 3000, 512, 64
    rng = np.random.default_rng(3)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.permutation(NU)[:B].astype(np.int32)                 # unique users, unique items ...
    perm = rng.permutation(NI)
    pid, nid = perm[:B].astype(np.int32), perm[B:2 * B].astype(np.int32)
    uid[1] = uid[0]                                                # ... but triplets 0 and 1 share their user row (referenced exactly twice)
    bad = nid.copy(); bad[1] = NI + 7                              # and triplet 1 carries an invalid negative
    with env(ORX_PAIR_ALWAYS=1, ORX_NO_PAIR=None):
        ctx = rt.Context(0)
        tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
        o = rt.Optimizer.sgd(0.05, ctx=ctx)
        with pytest.raises(IndexError):          # (the reference's CPU gather raises on an out-of-range id)
            rt.pairwise_step("bpr", o, tU, tV, tb, uid, pid, bad)
        got = tU.read()
    # what triplet 0 alone does to the shared user row: the loss is a mean over B triplets (g = -sigma(-x) / B), l2_loss a sum
    from oracle import numpy_oracle as orc
    g1 = orc.bpr_grads(U, V, b, uid[:1], pid[:1], nid[:1])               # one triplet, mean over 1: gu = g (p - n) + u
    loss_part = (g1["gu"][0] - U[uid[0]]) / B
    want_row = U[uid[0]] - 0.05 * (loss_part + U[uid[0]])
    assert np.abs(got[uid[0]] - want_row).max() < 1e-6, "the valid triplet's update of the shared row was lost"


# ---------------------------------------------------------------------------------------------- pointwise steps (round 6)
def _point_case(seed, NU, NI, B, D, K):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32); w = rng.uniform(-.3, .3, (D, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); iid = rng.integers(0, NI, (K, B)).astype(np.int32)
    lab = (rng.uniform(size=(K, B)) < 0.4).astype(np.float32)
    return U, V, b, w, uid, iid, lab


def _point_run(model, U, V, b, w, uid, iid, lab, sigmoid=False):
    rt = _rt()
    ctx = rt.Context(0)
    tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
    tw = rt.Table(*w.shape, ctx).write(w) if model == "gmf" else None
    o = rt.Optimizer.sgd(0.05, ctx=ctx)
    K, B = uid.shape
    l, l2 = rt.pointwise_step(model, o, tU, tV, tb, tw, uid.reshape(-1), iid.reshape(-1), lab.reshape(-1), K=K, B=B, a=1.5, b_w=0.7, sigmoid=sigmoid)
    out = dict(U=tU.read(), V=tV.read(), b=tb.read(), loss=np.asarray(l, np.float64), l2=np.asarray(l2, np.float64),
               pairs=ctx.stat("pairs"), max_dup=ctx.stat("max_dup"))
    if model == "gmf":
        out["w"] = tw.read()
    return out


def _point_oracle(model, U, V, b, w, uid, iid, lab, sigmoid=False):
    from oracle import numpy_oracle as orc
    U, V, b, w = U.copy(), V.copy(), b.copy(), w.copy()
    opt = orc.SGD(lr=0.05)
    ls = []
    for k in range(uid.shape[0]):
        if model == "gmf":
            ls.append(orc.gmf_step(U, V, b, w, uid[k], iid[k], lab[k], opt))
        else:
            ls.append(orc.wrmf_step(U, V, b, uid[k], iid[k], lab[k], opt, a=1.5, b_w=0.7, sigmoid=sigmoid))
    out = dict(U=U, V=V, b=b, loss=np.array([x[0] for x in ls], np.float64), l2=np.array([x[1] for x in ls], np.float64))
    if model == "gmf":
        out["w"] = w
    return out


@pytest.mark.parametrize("D", [16, 32, 64, 128])
@pytest.mark.parametrize("model,sigmoid", [("gmf", False), ("wrmf", False), ("wrmf", True)])
@pytest.mark.parametrize("B", [4096, 4095])
def test_pointwise_paired_step_equals_unpaired_step_and_oracle(D, model, sigmoid, B):
    """GMF / WRMF (gmf.py:22-34, wrmf.py:21-34, pointwise_mse_loss.py:18-31) with the pairing plan: the two samples of a row referenced
    exactly twice exchange their gradients in a wavefront and one of them writes the row; a sample's LABEL travels with its ids in the
    plan's input records (word z) when the plan moves it next to its partner.  Paired = unpaired = oracle, and pairs were accepted."""
    U, V, b, w, uid, iid, lab = _point_case(31 + D, 30000, 30000, B, D, K=5)
    with env(ORX_PAIR_ALWAYS=1, ORX_NO_PAIR=None, ORX_POINT_NO_PAIR=None):
        paired = _point_run(model, U, V, b, w, uid, iid, lab, sigmoid)
    with env(ORX_POINT_NO_PAIR=1):
        plain = _point_run(model, U, V, b, w, uid, iid, lab, sigmoid)
    want = _point_oracle(model, U, V, b, w, uid, iid, lab, sigmoid)
    assert paired["pairs"] > 5 * 50, "pairing was idle: %d accepted pairs" % paired["pairs"]
    assert plain["pairs"] == 0
    assert paired["max_dup"] < plain["max_dup"]
    keys = ("U", "V", "b", "loss", "l2") + (("w",) if model == "gmf" else ())
    for what, a, c in (("paired vs oracle", paired, want), ("unpaired vs oracle", plain, want), ("paired vs unpaired", paired, plain)):
        for k in keys:
            assert rel_err(a[k], c[k]) < TOL, (what, k, rel_err(a[k], c[k]))


def test_pointwise_pairing_in_a_long_call_with_an_invalid_id_and_skew():
    """K = 40 (the in-launch apply across many steps, dead list entries), one sample with an out-of-range item id (skipped, never
    anybody's partner), a hot item (hundreds of references: the deposit / staging path beside the pairs)"""
    rt = _rt()
    U, V, b, w, uid, iid, lab = _point_case(77, 50000, 60000, 8192, 64, K=40)
    iid[:, 100:400] = 12345
    bad = iid.copy(); bad[3, 17] = 60000 + 5
    with env(ORX_PAIR_ALWAYS=1, ORX_NO_PAIR=None, ORX_POINT_NO_PAIR=None):
        ctx = rt.Context(0)
        tU, tV, tb = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b))
        o = rt.Optimizer.sgd(0.0005, ctx=ctx)
        import torch
        du, di, dl = (torch.from_numpy(x.reshape(-1)).cuda() for x in (uid, bad, lab))     # (device ids: an invalid id is reported, not raised)
        l, l2 = rt.pointwise_step("wrmf", o, tU, tV, tb, None, du, di, dl, K=40, B=8192, a=1.5, b_w=0.7)
        assert ctx.stat("pairs") > 40 * 100
        with pytest.raises(Exception):
            ctx.check_index_error()
        got = dict(U=tU.read(), V=tV.read(), b=tb.read())
    from oracle import numpy_oracle as orc
    Uo, Vo, bo = U.copy(), V.copy(), b.copy()
    opt = orc.SGD(lr=0.0005)
    for k in range(40):
        keep = np.ones(8192, bool)
        if k == 3:
            keep[17] = False                                     # the reference's gather raises on it; the device skips the sample
        # (WRMF's loss is a SUM over the batch, pointwise_mse_loss.py:31: the other samples' gradients do not notice the missing one)
        orc.wrmf_step(Uo, Vo, bo, uid[k][keep], iid[k][keep], lab[k][keep], opt, a=1.5, b_w=0.7)
    for k, want in (("U", Uo), ("V", Vo), ("b", bo)):
        assert rel_err(got[k], want) < TOL, (k, rel_err(got[k], want))
