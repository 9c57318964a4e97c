"""The HIP path against the fixtures minted from the REFERENCE'S OWN model text (tests/golden/refstub/, written by
tests/golden/make_golden_tf.py --backend stub: openrec.tf2.recommenders.{BPR, UCML} and openrec.tf2.metrics.ranking_metrics
imported from the reference tree and driven by the train step of tf2_examples/bpr_citeulike.py:33-39).

tests/test_reference_goldens.py holds the ORACLE to these files on the CPU; here the same files go to the device through the
C ABI -- no oracle in between:
  * ucmlc_*: UCML with censor_vec after every step (ucml.py:44-48; latent_factor.py:17-23), SGD / Adagrad / Adam;
  * large_bpr_*: BPR at B = 1024, N = 4096, D in {50, 64, 128}, stored rows + checksums of the rest;
  * metrics_s0: AUC / NDCG / Recall (metrics/ranking_metrics.py:8-69).
Tolerances as tests/test_gpu_pairwise.py::test_golden_fixtures: 1e-5 relative (conftest.TOL), Adam conftest.TOL_ADAM."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, OPT_KW, TOL, TOL_ADAM, rel_err

pytestmark = pytest.mark.gpu

REFSTUB = os.path.join(GOLDEN, "refstub")


def _files(prefix):
    return sorted(f for f in os.listdir(REFSTUB) if f.startswith(prefix) and f.endswith(".npz")) if os.path.isdir(REFSTUB) else []


def _make_opt(rt, kind):
    kw = OPT_KW[kind]
    if kind == "sgd":
        return rt.Optimizer.sgd(kw["lr"])
    if kind == "adagrad":
        return rt.Optimizer.adagrad(kw["lr"], kw["initial_accumulator_value"], kw["epsilon"])
    return rt.Optimizer.adam(kw["lr"], kw["beta_1"], kw["beta_2"], kw["epsilon"])


def test_the_fixtures_are_there():
    assert len(_files("ucmlc_")) >= 3 and len(_files("large_bpr_")) >= 9 and os.path.exists(os.path.join(REFSTUB, "metrics_s0.npz"))


@pytest.mark.parametrize("fused_censor", [True, False])
@pytest.mark.parametrize("fname", _files("ucmlc_"))
def test_ucml_with_censor_on_the_device_matches_the_reference_text(fname, fused_censor):
    """fused_censor: ORX_CENSOR inside the step's write-back; otherwise LatentFactor.censor as three separate calls in the
    reference's order (users, p-items, n-items)"""
    from openrec_amd import runtime as rt
    g = dict(np.load(os.path.join(REFSTUB, fname)))
    optkind = fname.split("_")[2]
    tol = TOL_ADAM if optkind == "adam" else TOL
    tU = rt.Table(*g["in_U"].shape).write(g["in_U"]); tV = rt.Table(*g["in_V"].shape).write(g["in_V"])
    tb = rt.Table(*g["in_b"].shape).write(g["in_b"])
    opt = _make_opt(rt, optkind)
    losses = []
    for s in range(int(g["steps"])):
        uid, pid, nid = np.roll(g["in_uid"], s), np.roll(g["in_pid"], 2 * s), np.roll(g["in_nid"], 3 * s)
        l, l2 = rt.pairwise_step("ucml", opt, tU, tV, tb, uid, pid, nid, margin=0.5, censor=fused_censor)
        if not fused_censor:
            tU.censor(uid); tV.censor(pid); tV.censor(nid)
        losses.append((l[0], l2[0]))
    assert rel_err(np.array(losses, np.float64), g["losses"]) < tol
    for k, t in (("U", tU), ("V", tV), ("b", tb)):
        assert rel_err(t.read(), g["out_" + k]) < tol, k


@pytest.mark.parametrize("fname", _files("large_bpr_"))
def test_bpr_at_b1024_n4096_on_the_device_matches_the_reference_text(fname):
    from openrec_amd import runtime as rt
    sys.path.insert(0, GOLDEN)
    try:
        import make_golden_tf                      # (inputs are regenerated from the recipe; nothing of the reference is touched)
    finally:
        sys.path.remove(GOLDEN)
    g = dict(np.load(os.path.join(REFSTUB, fname)))
    seed, D = (int(v) for v in g["recipe"])
    inp = make_golden_tf.large_inputs(seed, D)
    for k in ("U", "V", "b"):                      # the inputs ARE the ones the reference text saw
        w = inp[k].astype(np.float64)
        assert abs(w.sum() - g["sum_in_" + k][0]) <= 1e-9 * g["sum_in_" + k][1] and abs(np.abs(w).sum() - g["sum_in_" + k][1]) <= 1e-9 * g["sum_in_" + k][1]
    tU = rt.Table(*inp["U"].shape).write(inp["U"]); tV = rt.Table(*inp["V"].shape).write(inp["V"]); tb = rt.Table(*inp["b"].shape).write(inp["b"])
    opt = _make_opt(rt, "sgd")
    losses = []
    for s in range(int(g["steps"])):
        l, l2 = rt.pairwise_step("bpr", opt, tU, tV, tb, np.roll(inp["uid"], s), np.roll(inp["pid"], 2 * s), np.roll(inp["nid"], 3 * s))
        losses.append((l[0], l2[0]))
    assert rel_err(np.array(losses, np.float64), g["losses"]) < TOL
    for k, t, sel in (("U", tU, g["sel_U"]), ("V", tV, g["sel_V"]), ("b", tb, g["sel_V"])):
        w = t.read()
        assert rel_err(w[sel], g["out_" + k]) < TOL, k
        a = np.abs(w.astype(np.float64)).sum()                                   # the rows that are not stored
        assert abs(a - g["sum_out_" + k][1]) <= TOL * g["sum_out_" + k][1], k


def test_device_metrics_match_the_reference_text():
    from openrec_amd import runtime as rt
    from openrec_amd.tf2 import metrics
    g = np.load(os.path.join(REFSTUB, "metrics_s0.npz"))
    at = [int(a) for a in g["at"]]
    pred, pos, excl = g["in_pred"].astype(np.float32), g["in_pos"], g["in_excl"]
    got = rt.rank_metrics(pos, excl, at, pred=pred)
    assert np.allclose(got["auc"], g["auc"], rtol=1e-6, atol=1e-7)
    assert np.allclose(got["ndcg"], g["ndcg"], rtol=1e-5, atol=1e-6)
    assert np.allclose(got["recall"], g["recall"], rtol=1e-6, atol=1e-7)
    csr = rt.rank_metrics_csr(rt.SparseMask.from_dense(pos), rt.SparseMask.from_dense(excl), at, pred=pred)
    for k in ("auc", "recall"):                                       # (both kernels count the same integers ...)
        assert np.array_equal(np.asarray(csr[k]), np.asarray(got[k])), k
    assert np.allclose(csr["ndcg"], got["ndcg"], rtol=1e-6, atol=0)   # (... only the order of the per-user NDCG sum may differ)
    assert np.allclose(csr["ndcg"], g["ndcg"], rtol=1e-5, atol=1e-6)
    # the reference-named functions of the package (openrec.tf2.metrics.ranking_metrics)
    assert np.allclose(metrics.AUC(pos, pred, excl), g["auc"], rtol=1e-6, atol=1e-7)
    assert np.allclose(metrics.NDCG(pos, pred, excl, at=at), g["ndcg"], rtol=1e-5, atol=1e-6)
    assert np.allclose(metrics.Recall(pos, pred, excl, at=at), g["recall"], rtol=1e-6, atol=1e-7)
