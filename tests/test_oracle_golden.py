"""Pin the NumPy oracle against the torch-autograd golden fixtures
(tests/golden/make_golden.py) and against analytic known answers
(SURVEY.md Appendix A.7).  CPU only."""
import numpy as np
import pytest

from conftest import golden_files, golden_source, load_golden, parse_case, rel_err, OPT_KW, TOL, TOL_ADAM
from oracle import numpy_oracle as orc


def make_opt(kind):
    return {"sgd": orc.SGD, "adagrad": orc.Adagrad, "adam": orc.AdamTFSparse}[kind](**OPT_KW[kind])


def run_oracle_case(g, model, optkind, dtype):
    U, V, b = (g["in_" + k].astype(dtype) for k in ("U", "V", "b"))
    w = g["in_w"].astype(dtype)
    opt = make_opt(optkind)
    losses = []
    for s in range(int(g["steps"])):
        uid, pid, nid = np.roll(g["in_uid"], s), np.roll(g["in_pid"], 2 * s), np.roll(g["in_nid"], 3 * s)
        lab = np.roll(g["in_label"], s)
        if model == "bpr":
            losses.append(orc.bpr_step(U, V, b, uid, pid, nid, opt))
        elif model == "ucml":
            losses.append(orc.ucml_step(U, V, b, uid, pid, nid, opt, margin=0.5, do_censor=False))
        elif model == "gmf":
            losses.append(orc.gmf_step(U, V, b, w, uid, pid, lab, opt))
        elif model == "wrmf":
            losses.append(orc.wrmf_step(U, V, b, uid, pid, lab, opt, a=2.0, b_w=0.5))
    return dict(U=U, V=V, b=b, w=w, losses=np.array(losses, np.float64), opt=opt)


@pytest.mark.parametrize("fname", [f for f in golden_files() if f.split("_")[0] in ("bpr", "ucml", "gmf", "wrmf")])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-7), (np.float32, 1e-5)])
def test_oracle_matches_torch_autograd(fname, dtype, tol):
    model, D, optkind, seed = parse_case(fname)
    g = load_golden(fname)
    if golden_source(fname) == "tf":        # a float32 TensorFlow run (tests/golden/make_golden_tf.py --backend tf): north_star's tolerance
        tol = TOL_ADAM if optkind == "adam" else TOL
    r = run_oracle_case(g, model, optkind, dtype)
    assert rel_err(r["losses"], g["losses"]) < tol
    for k in ("U", "V", "b") + (("w",) if model == "gmf" else ()):
        assert rel_err(r[k], g["out_" + k]) < tol, k
    if optkind == "adagrad":
        assert rel_err(r["opt"].acc["U"], g["slot_U_acc"]) < tol
        assert rel_err(r["opt"].acc["V"], g["slot_V_acc"]) < tol
    if optkind == "adam":
        assert rel_err(r["opt"].m["V"], g["slot_V_m"]) < tol
        assert rel_err(r["opt"].v["V"], g["slot_V_v"]) < 10 * tol


# ----------------------------- known-answer tests (A.7) --------------------
def test_kat_all_zero_tables():
    B, NU, NI, D = 8, 4, 6, 16
    U, V, b = np.zeros((NU, D), np.float32), np.zeros((NI, D), np.float32), np.zeros((NI, 1), np.float32)
    uid = np.arange(B, dtype=np.int32) % NU
    pid = np.array([0, 1, 2, 3, 0, 1, 2, 3], np.int32)
    nid = np.array([4, 5, 4, 5, 4, 5, 4, 5], np.int32)
    loss, l2, x = orc.bpr_forward(U, V, b, uid, pid, nid)
    assert abs(loss - np.log(2)) < 1e-7 and l2 == 0 and (x == 0).all()
    gr = orc.bpr_grads(U, V, b, uid, pid, nid)
    assert np.allclose(gr["gbp"], -0.5 / B) and np.allclose(gr["gbn"], 0.5 / B)
    assert (gr["gu"] == 0).all() and (gr["gp"] == 0).all()


def test_kat_clamp_at_minus_30():
    D = 4
    U = np.array([[1, 0, 0, 0]], np.float32)
    V = np.array([[-40, 0, 0, 0], [0, 0, 0, 0], [-30, 0, 0, 0]], np.float32)
    b = np.zeros((3, 1), np.float32)
    # x = -40  -> clamped: loss term = softplus(30) = 30, loss-gradient exactly 0
    loss, _, x = orc.bpr_forward(U, V, b, np.array([0]), np.array([0]), np.array([1]))
    assert x[0] == -40 and abs(loss - 30.0) < 1e-5
    gr = orc.bpr_grads(U, V, b, np.array([0]), np.array([0]), np.array([1]))
    assert gr["g"][0] == 0 and np.array_equal(gr["gu"][0], U[0])
    # x = -30 exactly -> gradient flows (>=)
    gr = orc.bpr_grads(U, V, b, np.array([0]), np.array([2]), np.array([1]))
    assert gr["g"][0] < 0 and abs(gr["g"][0] + 1.0) < 1e-6


def test_kat_ucml_tie_and_equal_rows():
    U = np.full((1, 8), 0.25, np.float32)
    V = np.full((2, 8), 0.25, np.float32)
    b = np.zeros((2, 1), np.float32)
    loss, l2, h = orc.ucml_forward(U, V, b, np.array([0]), np.array([0]), np.array([1]), margin=0.5)
    assert h[0] == 0.5 and loss == 0.5
    gr = orc.ucml_grads(U, V, b, np.array([0]), np.array([0]), np.array([1]), margin=0.5)
    assert gr["g"][0] == 1 and np.array_equal(gr["gu"][0], U[0])   # u==p==n: only the L2 term
    # tie: margin - diff == 0 is ACTIVE
    b[0, 0] = 0.5
    gr = orc.ucml_grads(U, V, b, np.array([0]), np.array([0]), np.array([1]), margin=0.5)
    assert gr["g"][0] == 1


def test_kat_duplicate_user_sgd_and_adagrad():
    rng = np.random.default_rng(0)
    D, B = 8, 32
    U0, V0, b0 = (rng.uniform(-.05, .05, s).astype(np.float32) for s in ((3, D), (50, D), (50, 1)))
    uid = np.full(B, 1, np.int32)
    pid = rng.permutation(50)[:B].astype(np.int32)
    nid = ((pid + 7) % 50).astype(np.int32)
    gr = orc.bpr_grads(U0, V0, b0, uid, pid, nid)
    U, V, b = U0.copy(), V0.copy(), b0.copy()
    orc.bpr_step(U, V, b, uid, pid, nid, orc.SGD(lr=0.1))
    assert np.allclose(U[1], U0[1] - 0.1 * gr["gu"].sum(0), rtol=1e-5)
    assert np.array_equal(U[0], U0[0]) and np.array_equal(U[2], U0[2])
    U, V, b = U0.copy(), V0.copy(), b0.copy()
    opt = orc.Adagrad(lr=0.1, initial_accumulator_value=0.1)
    orc.bpr_step(U, V, b, uid, pid, nid, opt)
    G = gr["gu"].sum(0)
    assert np.allclose(opt.acc["U"][1], 0.1 + G * G, rtol=1e-5)          # (sum g)^2, not sum g^2


def test_kat_p_equals_n():
    rng = np.random.default_rng(1)
    U, V, b = (rng.uniform(-.05, .05, s).astype(np.float32) for s in ((2, 8), (3, 8), (3, 1)))
    uid, pid, nid = np.array([0]), np.array([2]), np.array([2])
    loss, _, x = orc.bpr_forward(U, V, b, uid, pid, nid)
    assert x[0] == 0
    V0 = V.copy()
    orc.bpr_step(U, V, b, uid, pid, nid, orc.SGD(lr=0.5))
    assert np.allclose(V[2], V0[2] - 0.5 * 2 * V0[2], atol=1e-7)        # +g u + p - g u + n


def test_kat_censor():
    W = np.zeros((4, 4), np.float32)
    W[0] = [2, 0, 0, 0]
    W[1] = [0.03, 0.04, 0, 0]          # norm 0.05 -> x10
    W[2] = [0.3, 0.4, 0, 0]
    order = orc.censor(W, np.array([2, 0, 2, 1, 0], np.int32))
    assert list(order) == [2, 0, 1]                                      # first-occurrence order
    assert np.allclose(W[0], [1, 0, 0, 0]) and np.allclose(W[1], [0.3, 0.4, 0, 0])
    assert np.allclose(np.linalg.norm(W[2]), 1.0) and (W[3] == 0).all()
