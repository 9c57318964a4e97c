"""The oracle against what the REFERENCE'S OWN model code computes (tests/golden/make_golden_tf.py).

* `test_fixtures_are_what_the_reference_text_computes` (build container only: needs /root/reference): the reference's classes
  -- imported, not copied -- run under the torch-backed stand-in `tensorflow` (tests/golden/tf_stub.py); every committed
  torch-autograd fixture of tests/golden/ must equal what they produce.  This is also the dry run of the TensorFlow script: same
  code path, same .npz schema, only the backend differs.
* tests/golden/refstub/*.npz (committed; minted by that dry run) hold the families the older fixtures lack -- UCML with
  `censor_vec`, the ranking metrics, B = 1024 / N = 4096 cases -- and are checked against the oracle everywhere.
* With a real TensorFlow 2.0.1, `make_golden_tf.py --backend tf` writes tests/golden/tf/*.npz and conftest.load_golden prefers them.
CPU only.
"""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, rel_err, OPT_KW
from oracle import numpy_oracle as orc
from oracle import metrics_oracle as mo

REF = "/root/reference"
REFSTUB = os.path.join(GOLDEN, "refstub")
sys.path.insert(0, GOLDEN)


def refstub_files(prefix):
    return sorted(f for f in os.listdir(REFSTUB) if f.startswith(prefix) and f.endswith(".npz")) if os.path.isdir(REFSTUB) else []


def make_opt(kind):
    return {"sgd": orc.SGD, "adagrad": orc.Adagrad, "adam": orc.AdamTFSparse}[kind](**OPT_KW[kind])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "openrec", "tf2")), reason="needs the reference tree (build container)")
def test_fixtures_are_what_the_reference_text_computes(tmp_path):
    import make_golden_tf
    import tf_stub
    try:
        written = make_golden_tf.main(["--backend", "stub", "--dtype", "float64", "--reference", REF, "--out", str(tmp_path)])
    finally:
        tf_stub.uninstall()
        if REF in sys.path:
            sys.path.remove(REF)
    assert len(written) == 21
    for fn in written:
        name = os.path.basename(fn)
        a, b = np.load(fn), np.load(os.path.join(GOLDEN, name))          # (the committed torch-autograd fixture, never tests/golden/tf)
        assert str(a["backend"]).startswith("stub") and str(a["dtype"]) == "float64"
        for k in b.files:
            assert k in a.files, (name, k)                                # the schema the GPU tests read
            x, y = np.asarray(a[k], np.float64), np.asarray(b[k], np.float64)
            assert x.shape == y.shape, (name, k, x.shape, y.shape)
            if k.startswith("in_") or k in ("dense", "sparse", "label", "steps"):
                assert np.array_equal(x, y), (name, k)
            else:
                assert rel_err(x, y) < 1e-7, (name, k, rel_err(x, y))     # fp64 runs stored as fp32


@pytest.mark.parametrize("fname", refstub_files("ucmlc_"))
@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-7), (np.float32, 1e-5)])
def test_oracle_ucml_with_censor_matches_the_reference_text(fname, dtype, tol):
    g = dict(np.load(os.path.join(REFSTUB, fname)))
    optkind = fname.split("_")[2]
    U, V, b = (g["in_" + k].astype(dtype) for k in ("U", "V", "b"))
    opt = make_opt(optkind)
    losses = []
    for s in range(int(g["steps"])):
        uid, pid, nid = np.roll(g["in_uid"], s), np.roll(g["in_pid"], 2 * s), np.roll(g["in_nid"], 3 * s)
        losses.append(orc.ucml_step(U, V, b, uid, pid, nid, opt, margin=0.5, do_censor=True))
    if optkind == "adam":
        tol = max(tol, 5e-5 if dtype == np.float32 else tol)
    assert rel_err(np.array(losses, np.float64), g["losses"]) < tol
    for k, w in (("U", U), ("V", V), ("b", b)):
        assert rel_err(w, g["out_" + k]) < tol, k


@pytest.mark.parametrize("fname", refstub_files("large_"))
def test_oracle_matches_the_reference_text_at_b1024_n4096(fname):
    import make_golden_tf
    g = dict(np.load(os.path.join(REFSTUB, fname)))
    seed, D = (int(v) for v in g["recipe"])
    inp = make_golden_tf.large_inputs(seed, D)
    for dtype, tol in ((np.float64, 2e-7), (np.float32, 1e-5)):
        U, V, b = (inp[k].astype(dtype) for k in ("U", "V", "b"))
        opt = orc.SGD(**OPT_KW["sgd"])
        losses = [orc.bpr_step(U, V, b, np.roll(inp["uid"], s), np.roll(inp["pid"], 2 * s), np.roll(inp["nid"], 3 * s), opt)
                  for s in range(int(g["steps"]))]
        assert rel_err(np.array(losses, np.float64), g["losses"]) < tol
        for k, w, sel in (("U", U, g["sel_U"]), ("V", V, g["sel_V"]), ("b", b, g["sel_V"])):
            assert rel_err(w[sel], g["out_" + k]) < tol, k
            s = np.array([w.astype(np.float64).sum(), np.abs(w.astype(np.float64)).sum()])      # the rows that are not stored
            assert abs(s[1] - g["sum_out_" + k][1]) <= tol * g["sum_out_" + k][1], k


def test_metrics_oracle_matches_the_reference_text():
    fn = os.path.join(REFSTUB, "metrics_s0.npz")
    if not os.path.exists(fn):
        pytest.skip("tests/golden/refstub/metrics_s0.npz not minted")
    g = np.load(fn)
    at = [int(a) for a in g["at"]]
    pred, pos, excl = g["in_pred"].astype(np.float64), g["in_pos"], g["in_excl"]
    assert np.allclose(mo.auc(pos, pred, excl), g["auc"], rtol=1e-6, atol=1e-9)
    assert np.allclose(mo.ndcg(pos, pred, excl, at=at), g["ndcg"], rtol=1e-6, atol=1e-9)
    assert np.allclose(mo.recall(pos, pred, excl, at=at), g["recall"], rtol=1e-6, atol=1e-9)


def test_the_tf_backend_refuses_to_pretend():
    """without TensorFlow the real backend must fail loudly, not fall back to the stand-in"""
    import make_golden_tf
    try:
        import tensorflow  # noqa: F401
        pytest.skip("a tensorflow module is importable here")
    except ImportError:
        pass
    with pytest.raises(ImportError):
        make_golden_tf.main(["--backend", "tf", "--out", "/tmp/should_not_exist_orx"])
