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


@pytest.mark.gpu
@pytest.mark.parametrize("NI,maxpos", [(1698, 8), (1000, 16), (4099, 40), (33, 3), (5000, 150), (70001, 20)])   # (several chunks of 63; 3 segments)
def test_csr_metrics_equal_the_mask_metrics(NI, maxpos):
    """orx_rank_metrics_csr (item lists -> device bitmaps) against the oracle and, bit for bit, against the byte-mask kernel:
    both count the same integers; only the order of the per-user NDCG sum (atomics in LDS) may differ."""
    from openrec_amd import runtime as rt
    from openrec_amd.tf2 import metrics
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(NI)
    n, D = 37, 24
    U = rng.normal(size=(200, D)).astype(np.float32) * 0.3; V = rng.normal(size=(NI, D)).astype(np.float32) * 0.3
    b = rng.normal(size=(NI, 1)).astype(np.float32) * 0.1
    uid = rng.integers(0, 200, n).astype(np.int32)
    pos = np.zeros((n, NI), bool)
    for q in range(n):
        pos[q, rng.choice(NI, rng.integers(1, maxpos + 1), replace=False)] = True
    pos[0] = False; pos[0, NI - 1] = True                                   # the last item (the bitmap's partial word)
    pos[2] = False                                                          # no positives: 0 / 0 -> NaN like TF
    excl = (rng.uniform(size=(n, NI)) < 0.05) & ~pos
    excl[1, np.nonzero(pos[1])[0][:1]] = True                               # a positive that is also excluded
    excl[3] = ~pos[3]                                                       # nothing left to rank against: AUC 0 / 0
    pred = orc.bpr_inference(U, V, b, uid).astype(np.float32)
    pred[4, :] = 0.25                                                       # a row of ties
    pred[5] = pred[5] * 1e-4                                                # scores so close that exp() rounds neighbours together
    pred[6] = 89.0 + np.abs(pred[6]) * 20                                   # exp overflows: every rank comparison is inf > inf
    pred[7] = -88.0 - np.abs(pred[7]) * 30                                  # exp underflows (to denormals and to zero)
    pred[8, ::2] = np.float32(0.5); pred[8, 1::2] = np.nextafter(np.float32(0.5), np.float32(1))      # one ulp apart
    at = [5, 50]
    sp, se = rt.SparseMask.from_dense(pos), rt.SparseMask.from_dense(excl)
    with np.errstate(all="ignore"):
        ref = dict(auc=mo.auc(pos, pred, excl), ndcg=mo.ndcg(pos, pred, excl, at), recall=mo.recall(pos, pred, excl, at))
    dense = rt.rank_metrics(pos, excl, at, pred=pred)
    got = rt.rank_metrics_csr(sp, se, at, pred=pred)
    for k in ("auc", "recall"):
        assert np.array_equal(got[k], dense[k], equal_nan=True), k
        assert np.allclose(got[k], ref[k], rtol=1e-5, atol=1e-6, equal_nan=True), k
    assert np.allclose(got["ndcg"], dense["ndcg"], rtol=1e-6, atol=0) and np.allclose(got["ndcg"], ref["ndcg"], rtol=1e-5, atol=1e-6)
    assert np.isnan(got["auc"][2]) and np.isnan(got["auc"][3]) and np.all(np.isnan(got["recall"][2]))
    # the reference-shaped functions take the lists too
    assert np.array_equal(metrics.AUC(sp, pred, se), got["auc"], equal_nan=True)
    assert np.array_equal(metrics.Recall(sp, pred, se, at=at), got["recall"], equal_nan=True)
    # scores computed and kept on the device
    tU = rt.Table(*U.shape).write(U); tV = rt.Table(*V.shape).write(V); tb = rt.Table(*b.shape).write(b)
    ds = rt.score_all_items("dot", tU, tV, tb, uid, device=True)
    assert isinstance(ds, rt.DeviceScores) and ds.shape == (n, NI)
    fused = rt.rank_metrics_csr(sp, se, at, kind="dot", user=tU, item=tV, bias=tb, uid=uid)
    kept = rt.rank_metrics_csr(sp, se, at, pred=ds)
    for k in ("auc", "ndcg", "recall"):
        assert np.array_equal(fused[k], kept[k], equal_nan=True) or np.allclose(fused[k], kept[k], rtol=1e-6, equal_nan=True)
    host = np.asarray(ds)                                                   # ... and the host copy on demand
    assert host.shape == (n, NI) and np.abs(host - orc.bpr_inference(U, V, b, uid)).max() < 1e-5
    again = rt.rank_metrics_csr(sp, se, at, pred=host)
    assert np.array_equal(again["auc"], kept["auc"], equal_nan=True)
    assert np.array_equal(metrics.AUC(sp, ds, se), kept["auc"], equal_nan=True)


@pytest.mark.gpu
def test_csr_metrics_reject_bad_lists():
    from openrec_amd import runtime as rt
    pred = np.random.default_rng(0).normal(size=(2, 100)).astype(np.float32)
    empty = rt.SparseMask.from_lists([[], []], 100)
    dup = rt.SparseMask(np.array([0, 2, 3]), np.array([5, 5, 7], np.int32), 100)         # built by hand: from_lists would dedup
    with pytest.raises(ValueError):
        rt.rank_metrics_csr(dup, empty, [10], pred=pred)
    oob = rt.SparseMask(np.array([0, 1, 2]), np.array([5, 100], np.int32), 100)
    with pytest.raises(IndexError):
        rt.rank_metrics_csr(oob, empty, [10], pred=pred)
    ok = rt.rank_metrics_csr(rt.SparseMask.from_lists([[5], [7]], 100), empty, [10], pred=pred)   # the context still works
    assert np.all(np.isfinite(ok["auc"]))


@pytest.mark.gpu
def test_evaluate_takes_the_dataset_batches():
    """eval_step of tf2_examples/bpr_citeulike.py:41-46 on Dataset.evaluation batches: the unmodified form (inference -> AUC /
    Recall, scores staying in device memory) and Recommender.evaluate, both against the oracle on the dense masks."""
    from openrec_amd.tf2.data import Dataset
    from openrec_amd.tf2.recommenders import BPR
    from openrec_amd.tf2.metrics import AUC, Recall
    from openrec_amd import runtime as rt
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(5)
    NU, NI = 300, 2500
    def raw(n):
        a = np.zeros(n, dtype=[("user_id", np.int32), ("item_id", np.int32)])
        a["user_id"] = rng.integers(0, NU, n); a["item_id"] = rng.integers(0, NI, n)
        return a
    train, val = Dataset(raw(6000), NU, NI, seed=1), Dataset(raw(900), NU, NI, seed=1)
    m = BPR(dim_user_embed=32, dim_item_embed=32, total_users=NU, total_items=NI)
    U, V, b = (m.user_latent_factor.table.read(), m.item_latent_factor.table.read(), m.item_bias.table.read())
    seen = 0
    for batch in val.evaluation(batch_size=64, excl_datasets=[train]):
        pred = m.inference(batch["user_id"])
        assert isinstance(pred, rt.DeviceScores)
        auc = AUC(pos_mask=batch["pos_mask"], pred=pred, excl_mask=batch["excl_mask"])
        rec = Recall(pos_mask=batch["pos_mask"], pred=pred, excl_mask=batch["excl_mask"], at=[50, 100])
        ev = m.evaluate(**batch, at=[50, 100])
        pos, excl = np.asarray(batch["pos_mask"]), np.asarray(batch["excl_mask"])
        ref_pred = orc.bpr_inference(U, V, b, batch["user_id"]).astype(np.float32)
        assert np.allclose(auc, mo.auc(pos, ref_pred, excl), rtol=2e-4, atol=2e-4)
        assert np.allclose(rec, mo.recall(pos, ref_pred, excl, [50, 100]), rtol=1e-3, atol=1e-3)
        assert np.array_equal(ev["auc"], auc) and np.array_equal(ev["recall"], rec)
        seen += len(batch["user_id"])
    assert seen == len(val.datastore.warm_users())


@pytest.mark.gpu
def test_csr_metrics_sliced_batch():
    """300 users: the fused call scores and sweeps the batch in slices on two streams; same numbers as one pass over kept scores."""
    from openrec_amd import runtime as rt
    rng = np.random.default_rng(11)
    n, NI, D = 300, 3000, 32
    U = rng.normal(size=(400, D)).astype(np.float32) * 0.3; V = rng.normal(size=(NI, D)).astype(np.float32) * 0.3
    b = rng.normal(size=(NI, 1)).astype(np.float32) * 0.1
    uid = rng.integers(0, 400, n).astype(np.int32)
    pos = rt.SparseMask.from_lists([rng.choice(NI, rng.integers(1, 12), replace=False) for _ in range(n)], NI)
    excl = rt.SparseMask.from_lists([rng.choice(NI, 50, replace=False) for _ in range(n)], NI)
    tU = rt.Table(*U.shape).write(U); tV = rt.Table(*V.shape).write(V); tb = rt.Table(*b.shape).write(b)
    kept = rt.rank_metrics_csr(pos, excl, [10, 100], pred=rt.score_all_items("dot", tU, tV, tb, uid, device=True))
    for _ in range(2):
        fused = rt.rank_metrics_csr(pos, excl, [10, 100], kind="dot", user=tU, item=tV, bias=tb, uid=uid)
        for k in ("auc", "ndcg", "recall"):
            assert np.array_equal(fused[k], kept[k]), k
    pd, ed = np.asarray(pos), np.asarray(excl)
    pred = np.asarray(rt.score_all_items("dot", tU, tV, tb, uid))
    assert np.allclose(kept["auc"], mo.auc(pd, pred, ed), rtol=1e-5, atol=1e-6)
