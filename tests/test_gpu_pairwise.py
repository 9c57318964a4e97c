"""GPU parity: the HIP train step (through the C ABI) against the NumPy oracle
and the committed torch-autograd golden fixtures.  Tolerance: 1e-5 relative on
loss and on every table (BASELINE.json north_star), bit-exact on WHICH rows
were touched."""
import numpy as np
import pytest

from conftest import golden_files, load_golden, parse_case, rel_err, OPT_KW

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _rt():
    from openrec_amd import runtime as rt
    return rt


def _make_opt(rt, kind):
    kw = OPT_KW[kind]
    if kind == "sgd":
        return rt.Optimizer.sgd(kw["lr"])
    if kind == "adagrad":
        return rt.Optimizer.adagrad(kw["lr"], kw["initial_accumulator_value"], kw["epsilon"])
    return rt.Optimizer.adam(kw["lr"], kw["beta_1"], kw["beta_2"], kw["epsilon"])


def _oracle_opt(kind):
    from oracle import numpy_oracle as orc
    return {"sgd": orc.SGD, "adagrad": orc.Adagrad, "adam": orc.AdamTFSparse}[kind](**OPT_KW[kind])


def _tables(rt, U, V, b):
    tU = rt.Table(*U.shape).write(U)
    tV = rt.Table(*V.shape).write(V)
    tb = rt.Table(*b.shape).write(b)
    return tU, tV, tb


@pytest.mark.parametrize("fname", golden_files("bpr") + golden_files("ucml"))
def test_golden_fixtures(fname):
    rt = _rt()
    model, D, optkind, seed = parse_case(fname)
    g = load_golden(fname)
    tU, tV, tb = _tables(rt, g["in_U"], g["in_V"], g["in_b"])
    opt = _make_opt(rt, optkind)
    losses = []
    for s in range(int(g["steps"])):
        uid, pid, nid = np.roll(g["in_uid"], s), np.roll(g["in_pid"], 2 * s), np.roll(g["in_nid"], 3 * s)
        l, l2 = rt.pairwise_step(model, opt, tU, tV, tb, uid, pid, nid, margin=0.5)
        losses.append((l[0], l2[0]))
    assert rel_err(np.array(losses), g["losses"]) < TOL
    assert rel_err(tU.read(), g["out_U"]) < TOL
    assert rel_err(tV.read(), g["out_V"]) < TOL
    assert rel_err(tb.read(), g["out_b"]) < TOL
    if optkind == "adagrad":
        assert rel_err(opt.slot(tU), g["slot_U_acc"]) < TOL
        assert rel_err(opt.slot(tb), g["slot_b_acc"]) < TOL
    if optkind == "adam":
        assert rel_err(opt.slot(tV, 0), g["slot_V_m"]) < TOL
        # v accumulates (1 - beta_2) * g^2 with beta_2 held in fp32 like TF's apply op (the hyper-parameter is cast to the
        # variable's dtype): 1 - fp32(0.999) = 1e-3 * (1 + 1.29e-5); the golden file was minted in fp64 with 1e-3 exactly
        assert rel_err(opt.slot(tV, 1), g["slot_V_v"]) < TOL + 1.3e-5


def _rand_case(seed, NU, NI, B, D, hot=True):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, B).astype(np.int32)
    pid = rng.integers(0, NI, B).astype(np.int32)
    nid = rng.integers(0, NI, B).astype(np.int32)
    if hot:
        uid[:16] = 3                             # a hot user (16x; lr * count stays < 1 so that
        #                                          three steps do not amplify fp32 rounding)
        nid[16:33] = pid[16:33]                  # p == n
        uid[-1], pid[-1], nid[-1] = NU - 1, NI - 1, 0            # boundary ids
    return U, V, b, uid, pid, nid


@pytest.mark.parametrize("model", ["bpr", "ucml"])
@pytest.mark.parametrize("optkind", ["sgd", "adagrad"])
@pytest.mark.parametrize("D", [16, 32, 50, 64, 128, 256])
def test_random_batches_vs_oracle(model, optkind, D):
    rt = _rt()
    from oracle import numpy_oracle as orc
    NU, NI, B = 3000, 5000, 4099                 # ragged tail (B not a multiple of anything)
    U, V, b, uid, pid, nid = _rand_case(11, NU, NI, B, D)
    tU, tV, tb = _tables(rt, U, V, b)
    opt = _make_opt(rt, optkind)
    oo = _oracle_opt(optkind)
    for s in range(3):
        u, p, n = np.roll(uid, 7 * s), np.roll(pid, 3 * s), np.roll(nid, s)
        l, l2 = rt.pairwise_step(model, opt, tU, tV, tb, u, p, n, margin=0.5)
        if model == "bpr":
            lr, l2r = orc.bpr_step(U, V, b, u, p, n, oo)
        else:
            lr, l2r = orc.ucml_step(U, V, b, u, p, n, oo, margin=0.5, do_censor=False)
        assert abs(l[0] - lr) <= TOL * abs(lr) and abs(l2[0] - l2r) <= TOL * abs(l2r)
    Ud, Vd, bd = tU.read(), tV.read(), tb.read()
    assert rel_err(Ud, U) < TOL and rel_err(Vd, V) < TOL and rel_err(bd, b) < TOL
    # index work is bit-exact: untouched rows are bit-identical to the start
    U0, V0, b0, *_ = _rand_case(11, NU, NI, B, D)
    touched_u = np.zeros(NU, bool); touched_i = np.zeros(NI, bool)
    for s in range(3):
        touched_u[np.roll(uid, 7 * s)] = True
        touched_i[np.roll(pid, 3 * s)] = True; touched_i[np.roll(nid, s)] = True
    assert np.array_equal(Ud[~touched_u], U0[~touched_u])
    assert np.array_equal(Vd[~touched_i], V0[~touched_i])
    assert (np.any(Ud != U0, axis=1) <= touched_u).all()


def test_multi_step_call_equals_single_steps():
    rt = _rt()
    from oracle import numpy_oracle as orc
    K, B, D = 5, 2048, 64
    U, V, b, *_ = _rand_case(3, 4000, 4000, B, D)
    rng = np.random.default_rng(9)
    uid = rng.integers(0, 4000, (K, B)).astype(np.int32)
    pid = rng.integers(0, 4000, (K, B)).astype(np.int32)
    nid = rng.integers(0, 4000, (K, B)).astype(np.int32)
    tU, tV, tb = _tables(rt, U, V, b)
    opt = rt.Optimizer.sgd(0.05)
    loss, l2 = rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid, K=K, B=B)
    oo = orc.SGD(lr=0.05)
    for s in range(K):
        lr, l2r = orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)
        assert abs(loss[s] - lr) <= TOL * abs(lr)
    assert rel_err(tU.read(), U) < TOL and rel_err(tV.read(), V) < TOL and rel_err(tb.read(), b) < TOL


def test_all_same_user_and_kats():
    rt = _rt()
    from oracle import numpy_oracle as orc
    B, D = 1024, 64
    U, V, b, uid, pid, nid = _rand_case(5, 10, 3000, B, D, hot=False)
    uid[:] = 4
    for optkind in ("sgd", "adagrad"):
        # 1024 gradients are summed into one row in an arbitrary (atomic) order: compare with the
        # fp64 oracle (truth) rather than with one particular fp32 summation order
        U1, V1, b1 = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
        tU, tV, tb = _tables(rt, U, V, b)
        opt = _make_opt(rt, optkind)
        rt.pairwise_step("bpr", opt, tU, tV, tb, uid, pid, nid)
        orc.bpr_step(U1, V1, b1, uid, pid, nid, _oracle_opt(optkind))
        assert rel_err(tU.read(), U1) < TOL and rel_err(tV.read(), V1) < TOL
    # all-zero tables: loss = ln 2, db[p] = -0.5/B * -lr ...
    tU = rt.Table(8, 64).fill(0); tV = rt.Table(16, 64).fill(0); tb = rt.Table(16, 1).fill(0)
    u = np.arange(8, dtype=np.int32); p = np.arange(8, dtype=np.int32); n = np.arange(8, 16, dtype=np.int32)
    l, l2 = rt.pairwise_step("bpr", rt.Optimizer.sgd(1.0), tU, tV, tb, u, p, n)
    assert abs(l[0] - np.log(2)) < 1e-6 and l2[0] == 0
    bb = tb.read()[:, 0]
    assert np.allclose(bb[:8], 0.5 / 8) and np.allclose(bb[8:], -0.5 / 8)
    assert (tU.read() == 0).all()


def test_out_of_range_id_raises():
    rt = _rt()
    tU = rt.Table(10, 64).fill(0); tV = rt.Table(10, 64).fill(0); tb = rt.Table(10, 1).fill(0)
    u = np.array([0, 1, 10], np.int32); p = np.array([0, 1, 2], np.int32); n = np.array([3, 4, 5], np.int32)
    with pytest.raises(IndexError):
        rt.pairwise_step("bpr", rt.Optimizer.sgd(0.1), tU, tV, tb, u, p, n)
    with pytest.raises(IndexError):
        tU.gather(np.array([-1], np.int32))


def test_loss_only_and_gather_and_censor():
    rt = _rt()
    from oracle import numpy_oracle as orc
    U, V, b, uid, pid, nid = _rand_case(8, 500, 700, 1000, 128)
    tU, tV, tb = _tables(rt, U, V, b)
    l, l2 = rt.pairwise_loss("ucml", tU, tV, tb, uid, pid, nid, margin=0.5)
    lr, l2r, _ = orc.ucml_forward(U, V, b, uid, pid, nid, 0.5)
    assert abs(l - lr) <= TOL * abs(lr) and abs(l2 - l2r) <= TOL * abs(l2r)
    assert np.array_equal(tU.read(), U)                      # forward only: tables untouched
    assert np.array_equal(tV.gather(pid), V[pid])            # Embedding gather is bit-exact
    W = np.random.default_rng(0).uniform(-1, 1, (500, 128)).astype(np.float32)
    W[7] *= 1e-3
    tW = rt.Table(500, 128).write(W)
    ids = np.concatenate([uid, [7, 7]]).astype(np.int32)
    tW.censor(ids, 0.1)
    orc.censor(W, ids, 0.1)
    assert rel_err(tW.read(), W) < 1e-6
    assert np.array_equal(tW.read()[~np.isin(np.arange(500), ids)], W[~np.isin(np.arange(500), ids)])


@pytest.mark.parametrize("D", [50, 128])
def test_ucml_censor_inside_multi_step_call(D):
    """ORX_CENSOR: UCML.censor_vec after every step of a K-step call (BASELINE configs[2])."""
    rt = _rt()
    from oracle import numpy_oracle as orc
    K, B = 4, 1500
    U, V, b, *_ = _rand_case(21, 900, 1100, B, D)
    U *= 30; V *= 30                                   # norms around 1: the censor really rescales
    rng = np.random.default_rng(5)
    uid = rng.integers(0, 900, (K, B)).astype(np.int32); pid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    nid = rng.integers(0, 1100, (K, B)).astype(np.int32)
    nid[:, :40] = pid[:, :40]                          # rows in both item lists are censored twice
    tU, tV, tb = _tables(rt, U, V, b)
    opt = rt.Optimizer.sgd(0.01)
    loss, l2 = rt.pairwise_step("ucml", opt, tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
    oo = orc.SGD(lr=0.01)
    for s in range(K):
        lr, _ = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
        assert abs(loss[s] - lr) <= 1e-5 * abs(lr)
    assert rel_err(tU.read(), U) < 1e-5 and rel_err(tV.read(), V) < 1e-5 and rel_err(tb.read(), b) < 1e-5


@pytest.mark.parametrize("D,optname,fallback", [(64, "sgd", "0"), (128, "sgd", "0"), (16, "adagrad", "0"), (128, "adagrad", "0"),
                                                (64, "sgd", "1"), (64, "sgd", "2"), (128, "adagrad", "4"), (256, "sgd", "0")])
def test_fused_censor_matches_censor_vec(D, optname, fallback, monkeypatch):
    """censor_vec fused into the write-back (rows referenced once) and into the duplicate apply (ucml.py:44-48,
    latent_factor.py:17-23): heavy duplication, rows shorter than min_norm (scaled x10 per censor), items
    referenced as positive and negative in the same step (censored twice).  fallback 4 = the separate passes."""
    monkeypatch.setenv("ORX_FORCE_FALLBACK", fallback)
    rt = _rt()
    from oracle import numpy_oracle as orc
    K, B, NU, NI = 6, 3000, 700, 900
    U, V, b, *_ = _rand_case(33, NU, NI, B, D)
    U *= 20; V *= 20
    V[::3] *= 1e-3; U[::5] *= 1e-3                     # norms far below 0.1: one censor = x10, two = x100
    rng = np.random.default_rng(9)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    nid[:, :30] = pid[:, :30]
    tU, tV, tb = _tables(rt, U, V, b)
    opt = rt.Optimizer.sgd(0.01) if optname == "sgd" else rt.Optimizer.adagrad(0.01)
    oo = orc.SGD(lr=0.01) if optname == "sgd" else orc.Adagrad(lr=0.01)
    loss, l2 = rt.pairwise_step("ucml", opt, tU, tV, tb, uid, pid, nid, K=K, B=B, margin=0.5, censor=True)
    for s in range(K):
        lr, l2r = orc.ucml_step(U, V, b, uid[s], pid[s], nid[s], oo, margin=0.5, do_censor=True)
        assert abs(loss[s] - lr) <= 1e-5 * abs(lr) and abs(l2[s] - l2r) <= 1e-5 * abs(l2r)
    assert rel_err(tU.read(), U) < 1e-5 and rel_err(tV.read(), V) < 1e-5 and rel_err(tb.read(), b) < 1e-5
