"""GPU parity of the fused pointwise (GMF / WRMF) step and of the all-item
scorer (Recommender.inference) against the oracle and the golden fixtures."""
import numpy as np
import pytest

from conftest import golden_files, load_golden, parse_case, rel_err, OPT_KW

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
    tol = TOL if optkind != "adam" else 5e-5      # Adam's m/(sqrt(v)+eps) amplifies rounding of tiny gradients
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
    for D in (50, 64):
        NU, NI = 300, 1000
        U = rng.normal(size=(NU, D)).astype(np.float32); V = rng.normal(size=(NI, D)).astype(np.float32)
        b = rng.normal(size=(NI, 1)).astype(np.float32); w = rng.normal(size=(D, 1)).astype(np.float32)
        tU = rt.Table(NU, D).write(U); tV = rt.Table(NI, D).write(V); tb = rt.Table(NI, 1).write(b); tw = rt.Table(D, 1).write(w)
        uid = rng.integers(0, NU, 37).astype(np.int32)
        assert rel_err(rt.score_all_items("dot", tU, tV, tb, uid), orc.bpr_inference(U, V, b, uid)) < TOL
        assert rel_err(rt.score_all_items("l2", tU, tV, tb, uid), orc.ucml_inference(U, V, b, uid)) < TOL
        ref = (U[uid][:, None, :] * V[None, :, :]) @ w[:, 0] + b[:, 0][None, :]
        assert rel_err(rt.score_all_items("gmf", tU, tV, tb, uid, w=tw), ref) < TOL
