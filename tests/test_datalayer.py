"""Index work, bit-exact against the reference's own data layer (run under a stub
`tensorflow` by tests/golden/make_golden_datalayer.py; fixture committed)."""
import numpy as np
import pytest

from conftest import load_golden


def _raw(g, u="raw_user", i="raw_item"):
    raw = np.zeros(len(g[u]), dtype=[("user_id", np.int32), ("item_id", np.int32)])
    raw["user_id"], raw["item_id"] = g[u], g[i]
    return raw


def _collect(it, n, keys):
    rows = []
    for batch in it:
        for k in range(len(batch[keys[0]])):
            rows.append([batch[key][k] for key in keys])
        if len(rows) >= n:
            break
    return np.array(rows[:n], np.float64)


def test_pairwise_sequence_matches_reference():
    from openrec_amd.tf2.data import Dataset
    g = load_golden("datalayer.npz")
    ds = Dataset(_raw(g), int(g["NU"]), int(g["NI"]), seed=7)
    got = _collect(ds.pairwise(batch_size=97), len(g["pair"]), ("user_id", "p_item_id", "n_item_id"))
    assert np.array_equal(got.astype(np.int64), g["pair"])
    b = next(Dataset(_raw(g), int(g["NU"]), int(g["NI"]), seed=7).pairwise(batch_size=50))
    assert b["user_id"].dtype == np.int32 and b["n_item_id"].shape == (50,)


def test_pointwise_sequences_match_reference():
    from openrec_amd.tf2.data import Dataset
    g = load_golden("datalayer.npz")
    ds = Dataset(_raw(g), int(g["NU"]), int(g["NI"]), seed=8)
    got = _collect(ds.stratified_pointwise(batch_size=64, pos_ratio=0.3), len(g["strat"]), ("user_id", "item_id", "label"))
    assert np.array_equal(got, g["strat"])
    ds = Dataset(_raw(g), int(g["NU"]), int(g["NI"]), seed=9)
    got = _collect(ds.per_pos_stratified_pointwise(batch_size=64, pos_ratio=0.2), len(g["perpos"]), ("user_id", "item_id", "label"))
    assert np.array_equal(got, g["perpos"])


def test_evaluation_masks_match_reference():
    from openrec_amd.tf2.data import Dataset
    g = load_golden("datalayer.npz")
    NU, NI = int(g["NU"]), int(g["NI"])
    train = Dataset(_raw(g, "raw2_user", "raw2_item"), NU, NI, seed=1)
    val = Dataset(_raw(g), NU, NI, seed=1)
    users, pos, excl = [], [], []
    for b in val.evaluation(batch_size=16, excl_datasets=[train]):
        users += list(b["user_id"]); pos.append(b["pos_mask"]); excl.append(b["excl_mask"])
    assert np.array_equal(np.array(users), g["eval_users"])
    assert np.array_equal(np.packbits(np.concatenate(pos), axis=1), g["eval_pos"])
    assert np.array_equal(np.packbits(np.concatenate(excl), axis=1), g["eval_excl"])


def test_take_and_parallel_producers():
    from openrec_amd.tf2.data import Dataset
    g = load_golden("datalayer.npz")
    ds = Dataset(_raw(g), int(g["NU"]), int(g["NI"]), seed=3)
    assert len(list(ds.pairwise(batch_size=10, take=5))) == 5
    batches = list(ds.pairwise(batch_size=32, num_parallel_calls=3, take=7))
    assert len(batches) == 7
    for b in batches:                                  # negatives are never positives of that user
        for u, n in zip(b["user_id"], b["n_item_id"]):
            assert not ds.datastore.is_positive(u, n)


def test_evaluation_masks_are_item_lists():
    """The masks of an evaluation batch are SparseMask objects: dense on demand, and the lists are what goes to the device."""
    from openrec_amd.tf2.data import Dataset
    from openrec_amd.runtime import SparseMask
    g = load_golden("datalayer.npz")
    NU, NI = int(g["NU"]), int(g["NI"])
    train = Dataset(_raw(g, "raw2_user", "raw2_item"), NU, NI, seed=1)
    val = Dataset(_raw(g), NU, NI, seed=1)
    b = next(val.evaluation(batch_size=16, excl_datasets=[train]))
    for k in ("pos_mask", "excl_mask"):
        m = b[k]
        assert isinstance(m, SparseMask) and m.shape == (16, NI) and m.dtype == bool
        d = np.asarray(m)
        assert d.dtype == bool and d.shape == (16, NI) and int(d.sum()) == m.items.size == int(m.sum())
        for q in range(16):
            r = m.row(q)
            assert np.array_equal(r, np.nonzero(d[q])[0]) and np.all(np.diff(r) > 0)      # sorted, distinct
        back = SparseMask.from_dense(d)
        assert np.array_equal(back.ptr, m.ptr) and np.array_equal(back.items, m.items)
    assert np.array_equal(b["pos_mask"][3], np.asarray(b["pos_mask"])[3]) and (~b["pos_mask"]).shape == (16, NI)
    with pytest.raises(IndexError):
        SparseMask.from_lists([[1, NI]], NI)
    rep = SparseMask.from_lists([[4, 2, 4, 2], []], 7)                                       # repeats collapse: a mask is a set
    assert np.array_equal(rep.items, [2, 4]) and np.array_equal(rep.ptr, [0, 2, 2])
