"""Ranking metrics: oracle known answers on CPU, device kernel vs oracle on the GPU."""
import numpy as np
import pytest

from oracle import metrics_oracle as mo


def test_metrics_oracle_known_answers():
    pred = np.array([[0.9, 0.1, 0.5, 0.3, 0.7]], np.float32)
    pos = np.array([[1, 0, 0, 1, 0]], bool)
    excl = np.array([[0, 0, 1, 0, 0]], bool)
    # eval items: 1 (0.1), 4 (0.7).  pos 0.9 beats both, pos 0.3 beats one -> AUC = 3 / 4
    assert mo.auc(pos, pred, excl)[0] == np.float32(0.75)
    # ranks (exp domain, excluded item scores 0): item0 rank 0; item3 (0.3): above = 0.9, 0.7 -> 2
    assert np.allclose(mo.recall(pos, pred, excl, at=[1, 3])[0], [0.5, 1.0])
    assert np.allclose(mo.ndcg(pos, pred, excl, at=[1, 3])[0], [1.0, 1.0 + 1.0 / np.log2(4.0)])


def _case(seed, n, NI, D):
    rng = np.random.default_rng(seed)
    U = rng.normal(size=(200, D)).astype(np.float32) * 0.3
    V = rng.normal(size=(NI, D)).astype(np.float32) * 0.3
    b = rng.normal(size=(NI, 1)).astype(np.float32) * 0.1
    uid = rng.integers(0, 200, n).astype(np.int32)
    pos = rng.uniform(size=(n, NI)) < 0.01
    pos[0] = False; pos[0, 5] = True                     # a user with a single positive
    excl = (rng.uniform(size=(n, NI)) < 0.02) & ~pos
    excl[1, np.nonzero(pos[1])[0][:1]] = True            # a positive that is also excluded
    return U, V, b, uid, pos, excl


@pytest.mark.gpu
@pytest.mark.parametrize("D", [50, 64])
def test_device_metrics_match_oracle(D):
    from openrec_amd import runtime as rt
    from openrec_amd.tf2 import metrics
    from oracle import numpy_oracle as orc
    U, V, b, uid, pos, excl = _case(D, 40, 1698, D)
    tU = rt.Table(*U.shape).write(U); tV = rt.Table(*V.shape).write(V); tb = rt.Table(*b.shape).write(b)
    pred = orc.bpr_inference(U, V, b, uid).astype(np.float32)
    at = [50, 100]
    ref = dict(auc=mo.auc(pos, pred, excl), ndcg=mo.ndcg(pos, pred, excl, at), recall=mo.recall(pos, pred, excl, at))
    # (1) reference-shaped functions on host scores
    assert np.allclose(metrics.AUC(pos, pred, excl), ref["auc"], rtol=1e-5, atol=1e-6)
    assert np.allclose(metrics.NDCG(pos, pred, excl, at=at), ref["ndcg"], rtol=1e-5, atol=1e-6)
    assert np.allclose(metrics.Recall(pos, pred, excl, at=at), ref["recall"], rtol=1e-5, atol=1e-6)
    # (2) fused: scores never leave the device
    got = rt.rank_metrics(pos, excl, at, kind="dot", user=tU, item=tV, bias=tb, uid=uid)
    assert np.allclose(got["auc"], ref["auc"], rtol=2e-4, atol=2e-4)       # scores differ in the last ulp -> rare rank ties
    assert np.allclose(got["recall"], ref["recall"], rtol=1e-3, atol=1e-3)
    assert np.allclose(got["ndcg"], ref["ndcg"], rtol=1e-3, atol=1e-3)   # (ranking_metrics.py:31-52; same rare rank ties)
    dm = metrics.DictMean({"AUC": [], "Recall": [2]})
    dm.update_state({"AUC": got["auc"], "Recall": got["recall"]})
    r = dm.result()
    assert r["Recall"].shape == (2,) and abs(r["AUC"] - got["auc"].mean()) < 1e-6
