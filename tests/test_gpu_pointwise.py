"""GPU parity of the fused pointwise (GMF / WRMF) step and of the all-item
scorer (Recommender.inference) against the oracle and the golden fixtures."""
import numpy as np
import pytest

from conftest import TOL_ADAM, golden_files, load_golden, parse_case, rel_err, OPT_KW

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _opt(rt, kind):
    kw = OPT_KW[kind]
    if kind == "sgd":
        return rt.Optimizer.sgd(kw["lr"])
    if kind == "adagrad":
        return rt.Optimizer.adagrad(kw["lr"], kw["initial_accumulator_value"], kw["epsilon"])
    return rt.Optimizer.adam(kw["lr"], kw["beta_1"], kw["beta_2"], kw["epsilon"])


@pytest.mark.parametrize("fname", golden_files("gmf") + golden_files("wrmf"))
def test_golden_fixtures(fname):
    from openrec_amd import runtime as rt
    model, D, optkind, seed = parse_case(fname)
    g = load_golden(fname)
    tU = rt.Table(*g["in_U"].shape).write(g["in_U"]); tV = rt.Table(*g["in_V"].shape).write(g["in_V"])
    tb = rt.Table(*g["in_b"].shape).write(g["in_b"])
    tw = rt.Table(D, 1).write(g["in_w"]) if model == "gmf" else None
    opt = _opt(rt, optkind)
    losses = []
    for s in range(int(g["steps"])):
        uid, iid, lab = np.roll(g["in_uid"], s), np.roll(g["in_pid"], 2 * s), np.roll(g["in_label"], s)
        l, l2 = rt.pointwise_step(model, opt, tU, tV, tb, tw, uid, iid, lab, a=2.0, b_w=0.5)
        losses.append((l[0], l2[0]))
    assert rel_err(np.array(losses), g["losses"]) < TOL
    assert rel_err(tU.read(), g["out_U"]) < TOL and rel_err(tV.read(), g["out_V"]) < TOL
    assert rel_err(tb.read(), g["out_b"]) < TOL
    if model == "gmf":
        assert rel_err(tw.read(), g["out_w"]) < TOL


@pytest.mark.parametrize("model", ["gmf", "wrmf"])
@pytest.mark.parametrize("optkind", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("D", [32, 50, 64, 128])
def test_random_batches_vs_oracle(model, optkind, D):
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(4)
    NU, NI, B = 1500, 2500, 3001
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32); w = rng.uniform(-.3, .3, (D, 1)).astype(np.float32)
    uid = rng.integers(0, NU, B).astype(np.int32); iid = rng.integers(0, NI, B).astype(np.int32)
    uid[:13] = 9; iid[5:20] = 77
    lab = (rng.uniform(size=B) < 0.4).astype(np.float32)
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    tw = rt.Table(D, 1).write(w) if model == "gmf" else None
    opt = _opt(rt, optkind)
    oo = {"sgd": orc.SGD, "adagrad": orc.Adagrad, "adam": orc.AdamTFSparse}[optkind](**OPT_KW[optkind])
    for s in range(3):
        u, i, y = np.roll(uid, 3 * s), np.roll(iid, s), np.roll(lab, 2 * s)
        l, l2 = rt.pointwise_step(model, opt, tU, tV, tb, tw, u, i, y, a=1.5, b_w=0.7)
        if model == "gmf":
            lr, l2r = orc.gmf_step(U, V, b, w, u, i, y, oo)
        else:
            lr, l2r = orc.wrmf_step(U, V, b, u, i, y, oo, a=1.5, b_w=0.7)
        assert abs(l[0] - lr) <= TOL * abs(lr) and abs(l2[0] - l2r) <= TOL * abs(l2r)
    tol = TOL if optkind != "adam" else TOL_ADAM      # (conftest.TOL_ADAM)
    assert rel_err(tU.read(), U) < tol and rel_err(tV.read(), V) < tol and rel_err(tb.read(), b) < tol
    if model == "gmf":
        assert rel_err(tw.read(), w) < tol
        fl, fl2 = rt.pointwise_loss("gmf", tU, tV, tb, tw, uid, iid, lab)
        rl, rl2, _ = orc.gmf_forward(U, V, b, w, uid, iid, lab)
        assert abs(fl - rl) <= TOL * abs(rl) and abs(fl2 - rl2) <= TOL * abs(rl2)


def test_inference_scores():
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(1)
    for D, NI, nq in ((50, 1000, 37), (64, 1000, 37), (128, 4133, 130), (256, 777, 65), (16, 90, 3), (64, 70000, 200)):
        NU = 300
        U = rng.normal(size=(NU, D)).astype(np.float32); V = rng.normal(size=(NI, D)).astype(np.float32)
        b = rng.normal(size=(NI, 1)).astype(np.float32); w = rng.normal(size=(D, 1)).astype(np.float32)
        tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b); tw = rt.Table(D, 1).write(w)
        uid = rng.integers(0, NU, nq).astype(np.int32)
        assert rel_err(rt.score_all_items("dot", tU, tV, tb, uid), orc.bpr_inference(U, V, b, uid)) < TOL
        assert rel_err(rt.score_all_items("l2", tU, tV, tb, uid), orc.ucml_inference(U, V, b, uid)) < TOL
        ref = (U[uid].astype(np.float64) * w[:, 0]) @ V.T.astype(np.float64) + b[:, 0][None, :]
        assert rel_err(rt.score_all_items("gmf", tU, tV, tb, uid, w=tw), ref) < TOL


@pytest.mark.parametrize("model,optkind,D,fallback", [("wrmf", "sgd", 64, "0"), ("gmf", "sgd", 128, "0"), ("wrmf", "adagrad", 32, "0"),
                                                      ("gmf", "sgd", 64, "8"), ("wrmf", "sgd", 64, "1")])
def test_skewed_items_and_hot_user(model, optkind, D, fallback, monkeypatch):
    """Items ~ Zipf(1.05) plus one hot user: rows referenced twice take the plain-store roles, hot rows the staging
    plan and the reduction tree (fallback 8: atomics instead of staging; 1: byte flags + atomics for every duplicate).
    fp64 oracle: a row that sums hundreds of gradients has no unique fp32 answer."""
    monkeypatch.setenv("ORX_FORCE_FALLBACK", fallback)
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    NU, NI, B, K = 9000, 5000, 8192, 4
    rng = np.random.default_rng(8)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    wk = rng.uniform(-.3, .3, (D, 1)).astype(np.float32)
    pw = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(pw / pw.sum()); perm = rng.permutation(NI)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); uid[:, :400] = 11
    iid = perm[np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1)].astype(np.int32)
    lab = (rng.random((K, B)) < 0.4).astype(np.float32)
    assert np.bincount(iid[0], minlength=NI).max() > 300
    tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b)
    tw = rt.Table(D, 1).write(wk) if model == "gmf" else None
    lr = 0.002
    opt = rt.Optimizer.sgd(lr) if optkind == "sgd" else rt.Optimizer.adagrad(lr)
    oo = orc.SGD(lr) if optkind == "sgd" else orc.Adagrad(lr)
    loss, l2 = rt.pointwise_step(model, opt, tU, tV, tb, tw, uid, iid, lab, K=K, B=B, a=2.0, b_w=0.5)
    U, V, b, wk = (x.astype(np.float64) for x in (U, V, b, wk))
    for s in range(K):
        if model == "gmf":
            lw, l2w = orc.gmf_step(U, V, b, wk, uid[s], iid[s], lab[s], oo)
        else:
            lw, l2w = orc.wrmf_step(U, V, b, uid[s], iid[s], lab[s], oo, a=2.0, b_w=0.5)
        assert abs(loss[s] - lw) <= 3e-5 * abs(lw) and abs(l2[s] - l2w) <= 3e-5 * abs(l2w), (s, loss[s], lw)
    tol = 2e-5 if fallback == "0" else 1e-4          # atomics add in arrival order
    for got, want in ((tU.read(), U), (tV.read(), V), (tb.read(), b)):
        assert np.abs(got - want).max() <= tol * np.abs(want).max()
    if model == "gmf":
        assert np.abs(tw.read() - wk).max() <= 1e-4 * np.abs(wk).max()


@pytest.mark.parametrize("model,NU,NI,B,K,D", [("gmf", 40000, 30000, 4096, 16, 64), ("wrmf", 3000, 2500, 8192, 10, 32),
                                               ("wrmf", 90000, 70000, 2048, 24, 128)])
def test_lazy_adam_is_the_dense_decay_adam(model, NU, NI, B, K, D, monkeypatch):
    """GMF / WRMF with TF-2.0 Adam: the K-step path applies the every-row-every-step rule lazily (rows replay their
    gradient-free steps when next referenced or read); two calls of K steps with a read in between, against the
    fp64 oracle's dense rule, and the whole-table-sweep form (ORX_ADAM_DENSE=1) against the same numbers."""
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(K + D)
    U32 = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V32 = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b32 = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32); w32 = rng.uniform(-.3, .3, (D, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (2 * K, B)).astype(np.int32); iid = rng.integers(0, NI, (2 * K, B)).astype(np.int32)
    lab = (rng.uniform(size=(2 * K, B)) < 0.4).astype(np.float32)
    U, V, b, w = (x.astype(np.float64) for x in (U32, V32, b32, w32))
    oo = orc.AdamTFSparse(0.002, 0.9, 0.999, 1e-7)
    ref_loss, ref_mid = [], None
    for s in range(2 * K):
        if model == "gmf":
            ref_loss.append(orc.gmf_step(U, V, b, w, uid[s], iid[s], lab[s], oo)[0])
        else:
            ref_loss.append(orc.wrmf_step(U, V, b, uid[s], iid[s], lab[s], oo, a=1.5, b_w=0.7)[0])
        if s == K - 1:
            ref_mid = U.copy()
    # fp32 against the fp64 oracle: Adam normalises every gradient to a step of ~lr, so a handful of rows whose
    # gradient is nearly zero part from the oracle by 1e-4 .. 5e-4 after 2K steps, in BOTH forms alike: all but
    # 0.1 % of the rows must be within 5e-5, none beyond 2e-3 -- and the two forms must agree with each other to 1e-5
    def close(dev, host):
        e = np.abs(np.asarray(dev, np.float64) - host).max(axis=1) / np.abs(host).max()
        return (e > 5e-5).mean() <= 1e-3 and e.max() < 2e-3
    got = {}
    for form in ("lazy", "dense"):
        if form == "dense":
            monkeypatch.setenv("ORX_ADAM_DENSE", "1")
        tU = rt.Table(NU, D).write(U32); tV = rt.Table(NI, D).write(V32); tb = rt.Table(NI, 1).write(b32)
        tw = rt.Table(D, 1).write(w32) if model == "gmf" else None
        opt = rt.Optimizer.adam(0.002, 0.9, 0.999, 1e-7)
        loss = []
        for rep in range(2):
            sl = slice(rep * K, (rep + 1) * K)
            loss += list(rt.pointwise_step(model, opt, tU, tV, tb, tw, uid[sl], iid[sl], lab[sl], K=K, B=B, a=1.5, b_w=0.7)[0])
            if rep == 0:
                assert close(tU.read(), ref_mid), form
        assert np.abs(np.array(loss) - np.array(ref_loss)).max() <= 2e-5 * np.abs(ref_loss).max(), form
        assert close(tU.read(), U) and close(tV.read(), V) and close(tb.read(), b), form
        assert close(opt.slot(tV, 0), oo.m["V"]) and rel_err(opt.slot(tV, 1), oo.v["V"]) < 2e-3, form
        if model == "gmf":
            assert rel_err(tw.read(), w) < 5e-5, form
        got[form] = (tU.read(), tV.read(), tb.read())
    for x, y in zip(got["lazy"], got["dense"]):       # (the same few rows amplify the replay's rounding)
        e = np.abs(x.astype(np.float64) - y).max(axis=1) / np.abs(y).max()
        assert (e > 1e-5).mean() <= 1e-3 and e.max() < 2e-3


# ---- GMF's Dense(1) gradient reduced and applied by the last workgroups of the step's own launch (kernels_pointwise.hip dense_tail) ----
def _gmf_run(optkind, U, V, b, w, uid, iid, lab, K):
    from openrec_amd import runtime as rt
    ctx = rt.Context(0)
    tU, tV, tb, tw = (rt.Table(*x.shape, ctx).write(x) for x in (U, V, b, w))
    kw = OPT_KW[optkind]
    opt = {"sgd": lambda: rt.Optimizer.sgd(kw["lr"], ctx=ctx),
           "adagrad": lambda: rt.Optimizer.adagrad(kw["lr"], kw["initial_accumulator_value"], kw["epsilon"], ctx=ctx),
           "adam": lambda: rt.Optimizer.adam(kw["lr"], kw["beta_1"], kw["beta_2"], kw["epsilon"], ctx=ctx)}[optkind]()
    B = uid.shape[1]
    l, l2 = rt.pointwise_step("gmf", opt, tU, tV, tb, tw, uid.reshape(-1), iid.reshape(-1), lab.reshape(-1), K=K, B=B)
    return dict(U=tU.read(), V=tV.read(), b=tb.read(), w=tw.read(), loss=np.asarray(l, np.float64), l2=np.asarray(l2, np.float64))


@pytest.mark.parametrize("optkind", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("D,B", [(64, 65536), (64, 4095), (16, 70001), (32, 1000), (128, 40000), (256, 9000), (64, 300000), (64, 7)])
def test_gmf_dense_gradient_inside_the_launch(optkind, D, B, monkeypatch):
    """gmf.py:26-32: the Dense(1, use_bias=False) kernel is a dense variable -- its gradient is the sum over the WHOLE batch (+ the l2 term)
    and the next step reads the updated kernel.  The float4 kernels sum it inside the step's launch: the last of 64 consecutive workgroups
    adds their rows, the last of those adds the group rows and applies the rule.  Held here: (1) to the oracle, (2) to the two reduce
    launches it replaced (ORX_POINT_NO_WTAIL=1), (3) to ITSELF run again, bit for bit -- which workgroup arrives last differs from run to
    run, what it adds in which order must not.  Sizes: one workgroup, one group, 64 groups (the benchmark's B), the capped grid of 16 384
    workgroups = 256 groups, a ragged last group."""
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(D + B)
    NU, NI, K = B + 1000, B + 777, 4
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32); w = rng.uniform(-.3, .3, (D, 1)).astype(np.float32)
    # every row referenced at most once per step: nothing but the Dense(1) sum is a sum over samples, so the whole step has to reproduce
    # bit for bit (rows with many references take atomics on the generic paths; their order is not what this test is about)
    uid = np.stack([rng.permutation(NU)[:B] for _ in range(K)]).astype(np.int32)
    iid = np.stack([rng.permutation(NI)[:B] for _ in range(K)]).astype(np.int32)
    lab = (rng.uniform(size=(K, B)) < 0.4).astype(np.float32)
    monkeypatch.delenv("ORX_POINT_NO_WTAIL", raising=False)
    got = _gmf_run(optkind, U, V, b, w, uid, iid, lab, K)
    again = _gmf_run(optkind, U, V, b, w, uid, iid, lab, K)
    monkeypatch.setenv("ORX_POINT_NO_WTAIL", "1")
    launches = _gmf_run(optkind, U, V, b, w, uid, iid, lab, K)
    monkeypatch.delenv("ORX_POINT_NO_WTAIL")
    for k in ("w", "U", "V", "b", "loss", "l2"):
        assert np.array_equal(got[k], again[k]), ("not reproducible", k)
    oo = {"sgd": orc.SGD, "adagrad": orc.Adagrad, "adam": orc.AdamTFSparse}[optkind](**OPT_KW[optkind])
    Uo, Vo, bo, wo = U.copy(), V.copy(), b.copy(), w.copy()
    ls = [orc.gmf_step(Uo, Vo, bo, wo, uid[k], iid[k], lab[k], oo) for k in range(K)]
    want = dict(U=Uo, V=Vo, b=bo, w=wo, loss=np.array([x[0] for x in ls]), l2=np.array([x[1] for x in ls]))
    tol = TOL if optkind != "adam" else TOL_ADAM
    for what, a, c in (("in-launch vs oracle", got, want), ("two launches vs oracle", launches, want), ("in-launch vs two launches", got, launches)):
        for k in ("w", "U", "V", "b", "loss", "l2"):
            assert rel_err(a[k], c[k]) < tol, (what, k, rel_err(a[k], c[k]))
    # the kernel MOVED (the test cannot pass on a gradient that never reached w)
    assert np.abs(got["w"] - w).max() > 0
