#!/usr/bin/env python3
"""Golden index-work fixtures from the REFERENCE'S OWN data layer.

openrec/tf2/data/{utils,dataset}.py are pure Python apart from `tf.constant` and
three dtype symbols, so they run here under a 5-symbol stub `tensorflow` module
(SURVEY.md E.6).  This script drives the reference generators in-process with a
seed and stores the sampled sequences (tests/golden/datalayer.npz).  It reads
/root/reference and therefore only runs in the build container; the fixture is
committed.

Run:  python tests/golden/make_golden_datalayer.py
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

stub = types.ModuleType("tensorflow")
stub.int32, stub.float32, stub.bool = np.int32, np.float32, np.bool_
stub.constant = lambda v, dtype=None: np.asarray(v, dtype=dtype)
sys.modules["tensorflow"] = stub
if not hasattr(np, "bool"):
    np.bool = bool            # dataset.py:66 uses np.bool (removed in NumPy >= 1.24)
sys.path.insert(0, REF)
from openrec.tf2.data.utils import _DataStore                       # noqa: E402
from openrec.tf2.data import dataset as ref_dataset                  # noqa: E402


def synthetic(seed, n_users, n_items, n_records):
    rng = np.random.default_rng(seed)
    raw = np.zeros(n_records, dtype=[("user_id", np.int32), ("item_id", np.int32)])
    raw["user_id"] = rng.integers(0, n_users, n_records)
    raw["item_id"] = rng.integers(0, n_items, n_records)
    return raw


def take(gen, n):
    out = []
    for _ in range(n):
        out.append(next(gen))
    return out


def main():
    NU, NI, NR = 60, 90, 700
    raw = synthetic(11, NU, NI, NR)
    out = dict(raw_user=raw["user_id"], raw_item=raw["item_id"], NU=NU, NI=NI)
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=7)
    s = take(ref_dataset._pairwise_generator(ds), 1800)           # > 2 epochs of 700 records
    out["pair"] = np.array([[d["user_id"], d["p_item_id"], d["n_item_id"]] for d in s], np.int64)
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=8)
    s = take(ref_dataset._stratified_pointwise_generator(ds, 0.3), 900)
    out["strat"] = np.array([[d["user_id"], d["item_id"], d["label"]] for d in s], np.float64)
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=9)
    s = take(ref_dataset._per_pos_stratified_pointwise_generator(ds, 0.2), 900)
    out["perpos"] = np.array([[d["user_id"], d["item_id"], d["label"]] for d in s], np.float64)
    # evaluation masks with an exclusion dataset
    raw2 = synthetic(12, NU, NI, 300)
    train = types.SimpleNamespace(datastore=_DataStore(raw_data=raw2, total_users=NU, total_items=NI, seed=1))
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=1)
    ev = list(ref_dataset._evaluation_generator(ds, [train]))
    out["eval_users"] = np.array([e["user_id"] for e in ev], np.int64)
    out["eval_pos"] = np.packbits(np.stack([e["pos_mask"] for e in ev]), axis=1)
    out["eval_excl"] = np.packbits(np.stack([e["excl_mask"] for e in ev]), axis=1)
    out["raw2_user"], out["raw2_item"] = raw2["user_id"], raw2["item_id"]
    np.savez_compressed(os.path.join(HERE, "datalayer.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()})


if __name__ == "__main__":
    main()
