"""Parity at BASELINE.json's full sizes (configs[1]: BPR D=64, 1M x 1M, B=65536; configs[2]: UCML D=128) against the
C/OpenMP oracle (oracle/orx_oracle.c, itself pinned to the NumPy oracle by tests/test_c_oracle.py), plus
size-independent properties of a train step: rows that no triplet references keep their exact bits, the update of
the item-bias table sums to zero for BPR (every triplet adds +g to one bias gradient and -g to another), and the
K-step call equals K single-step calls."""
import numpy as np
import pytest

from conftest import TOL, TOL_ADAM, delta_check

pytestmark = pytest.mark.gpu


def _tables(NU, NI, D, seed):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    return U, V, b


@pytest.mark.parametrize("model,D,optname,censor", [("bpr", 64, "sgd", False), ("ucml", 128, "sgd", False), ("bpr", 64, "adagrad", False),
                                                    ("ucml", 128, "sgd", True)])
def test_full_size_steps_match_the_c_oracle(model, D, optname, censor):
    """censor=True is BASELINE configs[2] as SURVEY.md 8(d) quotes it: UCML D=128 with censor_vec after every step (ucml.py:44-48:
    LatentFactor.censor on the user ids, the p-item ids, the n-item ids; latent_factor.py:17-23) -- on the device the censor is
    fused into the write-back of the step's rows, items referenced as positive AND negative are censored twice."""
    from openrec_amd import runtime as rt
    from oracle import c_oracle
    NU = NI = 1_000_000
    B, K = 65536, 3
    U, V, b = _tables(NU, NI, D, 1)
    if censor:      # rows on both sides of min_norm = 0.1: U(-.05, .05)^128 has norm 0.33 (censored to 1), a tenth of the rows stays below 0.1
        U[::10] *= 0.2; V[::10] *= 0.2
    rng = np.random.default_rng(2)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    opt = rt.Optimizer.sgd(0.05) if optname == "sgd" else rt.Optimizer.adagrad(0.05)
    U0, V0, b0 = U.copy(), V.copy(), b.copy()
    loss, l2 = rt.pairwise_step(model, opt, tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=censor)
    cpu = c_oracle.PairwiseCPU(model, optname, U, V, b, lr=0.05)
    for s in range(K):
        lw, l2w = cpu.step(uid[s], pid[s], nid[s])
        assert abs(loss[s] - lw) <= 1e-5 * abs(lw) and abs(l2[s] - l2w) <= 1e-5 * abs(l2w)
        if censor:                                            # ucml.py:46-48, in the reference's order
            c_oracle.censor(U, uid[s]); c_oracle.censor(V, pid[s]); c_oracle.censor(V, nid[s])
    gU, gV, gb = tU.read(), tV.read(), tb.read()
    for got, want in ((gU, U), (gV, V), (gb, b)):
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max()
    # the UPDATE of every element (a bound on the table cannot see the loss gradient at this batch size: conftest.delta_check)
    for name, w0, got, want in (("user", U0, gU, U), ("item", V0, gV, V), ("item_bias", b0, gb, b)):
        coef = delta_check(w0, got, want, steps=K * (3 if censor else 1), what=f"{model} {optname} {name}")   # (a censor = a norm + a divide: two more roundings)
        assert abs(coef - 1.0) <= 1e-4, (name, coef)
    # rows outside every id list keep their exact bits
    untouched_u = np.ones(NU, bool); untouched_u[uid.reshape(-1)] = False
    untouched_i = np.ones(NI, bool); untouched_i[pid.reshape(-1)] = False; untouched_i[nid.reshape(-1)] = False
    assert untouched_u.sum() > 0.7 * NU and untouched_i.sum() > 0.5 * NI
    assert np.array_equal(gU[untouched_u], U0[untouched_u]) and np.array_equal(gV[untouched_i], V0[untouched_i])
    assert np.array_equal(gb[untouched_i], b0[untouched_i])
    if model == "bpr" and optname == "sgd":
        # d loss / d b_p = -d loss / d b_n for every triplet, and bias takes no l2: the SGD update of the table sums to 0
        delta = (gb.astype(np.float64) - b0).sum()
        assert abs(delta) <= 1e-4 * np.abs(gb.astype(np.float64) - b0).sum()


@pytest.mark.parametrize("model,D", [("bpr", 64), ("ucml", 128)])
def test_full_size_loss_gradient_alone(model, D):
    """ORX_NO_L2 (tape over `loss` alone) at configs[1] / configs[2] sizes: the whole update IS the loss gradient
    (pairwise_log_loss.py:32 / ucml.py:39 through the gathers), nothing of the 100x larger l2 term hides it.  One step,
    every element's change against the C oracle, and the update's scale to 1e-3."""
    from openrec_amd import runtime as rt
    from oracle import c_oracle
    NU = NI = 1_000_000
    B = 65536
    U, V, b = _tables(NU, NI, D, 7)
    rng = np.random.default_rng(8)
    uid = rng.integers(0, NU, B).astype(np.int32); pid = rng.integers(0, NI, B).astype(np.int32); nid = rng.integers(0, NI, B).astype(np.int32)
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    U0, V0, b0 = U.copy(), V.copy(), b.copy()
    loss, l2 = rt.pairwise_step(model, rt.Optimizer.sgd(0.05), tU, tV, tb, uid, pid, nid, K=1, B=B, margin=0.5, no_l2=True)
    cpu = c_oracle.PairwiseCPU(model, "sgd", U, V, b, lr=0.05, l2w=0.0)
    lw, l2w = cpu.step(uid, pid, nid)
    assert abs(loss[0] - lw) <= 1e-5 * abs(lw) and abs(l2[0] - l2w) <= 1e-5 * abs(l2w)
    for name, w0, got, want in (("user", U0, tU.read(), U), ("item", V0, tV.read(), V), ("item_bias", b0, tb.read(), b)):
        assert np.abs(want - w0).max() > 0, name
        coef = delta_check(w0, got, want, steps=1, what=f"{model} no_l2 {name}")
        assert abs(coef - 1.0) <= 1e-3, (name, coef)


def test_k_step_call_equals_single_step_calls_at_full_size():
    """the K-step path (duplicate apply inside the next launch, ready-flag hand-off) and K one-step calls
    (separate apply launches) must leave the same tables"""
    from openrec_amd import runtime as rt
    NU = NI = 1_000_000
    B, K, D = 65536, 6, 64
    rng = np.random.default_rng(5)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    res = []
    for split in (False, True):
        tU = rt.Table(NU, D).init_uniform(seed=1); tV = rt.Table(NI, D).init_uniform(seed=2); tb = rt.Table(NI, 1).init_uniform(seed=3)
        opt = rt.Optimizer.sgd(0.05)
        if split:
            ls = [rt.pairwise_step("bpr", opt, tU, tV, tb, uid[s], pid[s], nid[s], K=1, B=B)[0][0] for s in range(K)]
        else:
            ls = list(rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)[0])
        res.append((tU.read(), tV.read(), tb.read(), np.array(ls)))
    for a, c in zip(res[0][:3], res[1][:3]):
        assert np.abs(a - c).max() <= 2e-7 * np.abs(c).max()          # only the fp32 order of >= 3-reference sums differs
    assert np.allclose(res[0][3], res[1][3], rtol=1e-6)


def test_full_size_lazy_adam_matches_the_c_oracles_dense_rule():
    """TF-2.0 Adam at configs[1] sizes: the lazily-applied rule (rows replay their gradient-free steps when next
    referenced; the final read flushes every row) against the C oracle, which sweeps the whole tables every step."""
    from openrec_amd import runtime as rt
    from oracle import c_oracle
    NU = NI = 1_000_000
    B, K, D = 65536, 8, 64
    U, V, b = _tables(NU, NI, D, 3)
    rng = np.random.default_rng(4)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    opt = rt.Optimizer.adam(0.002)
    U0 = U.copy()
    loss, l2 = rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
    cpu = c_oracle.PairwiseCPU("bpr", "adam", U, V, b, lr=0.002)
    for s in range(K):
        lw, l2w = cpu.step(uid[s], pid[s], nid[s])
        assert abs(loss[s] - lw) <= 1e-5 * abs(lw) and abs(l2[s] - l2w) <= 1e-5 * abs(l2w)
    gU, gV, gb = tU.read(), tV.read(), tb.read()
    for got, want in ((gU, U), (gV, V), (gb, b)):
        assert np.abs(got - want).max() <= TOL_ADAM * np.abs(want).max()
    assert np.abs(opt.slot(tV, 0) - cpu.m[1]).max() <= TOL * np.abs(cpu.m[1]).max()
    untouched_u = np.ones(NU, bool); untouched_u[uid.reshape(-1)] = False      # m = v = 0 there: the rule moves nothing
    assert untouched_u.sum() > 0.5 * NU and np.array_equal(gU[untouched_u], U0[untouched_u])


@pytest.mark.parametrize("model,D,censor", [("bpr", 64, False), ("ucml", 128, True)])
def test_the_form_the_driver_times_matches_the_c_oracle(model, D, censor):
    """bench.py's protocol at configs[1] / configs[2] sizes, held to the C oracle on EVERY element's update: one context, the
    per-call buffers reserved, ids resident in HBM, two warm-up calls, then ONE K = 20 call with no loss fetch -- a LATER call
    of the context, which plans without the read-back (staging off, rows referenced >= 3 times on fp32 atomics), with pairing on.
    The other full-size tests make one call of a fresh shape each, which takes the read-back form."""
    import torch
    from openrec_amd import runtime as rt
    from oracle import c_oracle
    NU = NI = 1_000_000
    B, W, K = 65536, 5, 20
    U, V, b = _tables(NU, NI, D, 11)
    if censor:
        U[::10] *= 0.2; V[::10] *= 0.2
    rng = np.random.default_rng(12)
    uid = rng.integers(0, NU, (W + K, B)).astype(np.int32); pid = rng.integers(0, NI, (W + K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (W + K, B)).astype(np.int32)
    ctx = rt.Context(0)
    tU = rt.Table(NU, D, ctx).write(U); tV = rt.Table(NI, D, ctx).write(V); tb = rt.Table(NI, 1, ctx).write(b)
    opt = rt.Optimizer.sgd(0.05, ctx=ctx)
    U0, V0, b0 = U.copy(), V.copy(), b.copy()
    du, dp, dn = (torch.from_numpy(x).cuda().contiguous() for x in (uid, pid, nid))
    torch.cuda.synchronize()
    rt.pairwise_reserve(opt, tU, tV, tb, K, B)
    w1 = W - W // 2
    losses = []
    for first, count in ((0, w1), (w1, W // 2)):                                    # bench.py: the warm-up goes in two calls
        l, l2 = rt.pairwise_step(model, opt, tU, tV, tb, du[first:first + count], dp[first:first + count], dn[first:first + count],
                                 K=count, B=B, margin=0.5, censor=censor)
        losses += list(zip(l, l2))
    nowait_before = ctx.stat("nowait_calls")
    assert ctx.stat("quiet") == 1, "uniform ids over 1M rows must leave quiet plan counters"
    r = rt.pairwise_step(model, opt, tU, tV, tb, du[W:], dp[W:], dn[W:], K=K, B=B, margin=0.5, censor=censor, want_loss=False)
    assert r is None
    ctx.synchronize()
    assert ctx.stat("nowait_calls") == nowait_before + 1, "the timed call of the driver's protocol must not wait for its plan's counters"
    assert ctx.stat("pairs") > 0, "pairing was idle in the timed call"
    cpu = c_oracle.PairwiseCPU(model, "sgd", U, V, b, lr=0.05)
    for s in range(W + K):
        lw, l2w = cpu.step(uid[s], pid[s], nid[s])
        if s < W:
            assert abs(losses[s][0] - lw) <= 1e-5 * abs(lw) and abs(losses[s][1] - l2w) <= 1e-5 * abs(l2w)
        if censor:
            c_oracle.censor(U, uid[s]); c_oracle.censor(V, pid[s]); c_oracle.censor(V, nid[s])
    gU, gV, gb = tU.read(), tV.read(), tb.read()
    for name, w0, got, want in (("user", U0, gU, U), ("item", V0, gV, V), ("item_bias", b0, gb, b)):
        assert np.abs(got - want).max() <= 1e-5 * np.abs(want).max(), name
        coef = delta_check(w0, got, want, steps=(W + K) * (3 if censor else 1), what=f"driver form {model} {name}")
        assert abs(coef - 1.0) <= 1e-4, (name, coef)
    untouched_u = np.ones(NU, bool); untouched_u[uid.reshape(-1)] = False
    assert untouched_u.sum() > 0.1 * NU and np.array_equal(gU[untouched_u], U0[untouched_u])
