#!/usr/bin/env python3
"""Distribution fixture for the on-device pairwise sampler, from the REFERENCE'S OWN generator.

`openrec/tf2/data/dataset.py:7-16` (_pairwise_generator) over `utils.py:82-87` (next_random_record: shuffle, pop) and
`utils.py:102-116` (sample_negative_items: uniform draws rejected while positive) is driven in-process, seeded, under the
same 5-symbol stub `tensorflow` as make_golden_datalayer.py, for E epochs over a small interaction set.  Stored
(tests/golden/sampler_hist.npz): the records, how often every (user, item) record was emitted as the positive, and the
histogram of negatives per user.  The device sampler is counter-based and cannot reproduce CPython's Mersenne-Twister
SEQUENCE; what it must reproduce is this DISTRIBUTION (tests/test_gpu_sampler.py: exact positive counts, chi-square on
the negatives).  Reads /root/reference: runs in the build container only; the fixture is committed.

Run:  python tests/golden/make_golden_sampler.py
"""
import os
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
    np.bool = bool
sys.path.insert(0, REF)
from openrec.tf2.data.utils import _DataStore                       # noqa: E402
from openrec.tf2.data import dataset as ref_dataset                  # noqa: E402


def main():
    NU, NI, NR, E = 40, 64, 400, 300
    rng = np.random.default_rng(5)
    raw = np.zeros(NR, dtype=[("user_id", np.int32), ("item_id", np.int32)])
    raw["user_id"] = rng.integers(0, NU, NR); raw["item_id"] = rng.integers(0, NI, NR)
    raw["user_id"][:30] = 7; raw["item_id"][:30] = np.arange(30)         # a user with almost half of the items positive
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=123)
    gen = ref_dataset._pairwise_generator(ds)
    pos = np.zeros((NU, NI), np.int32); neg = np.zeros((NU, NI), np.int32)
    for _ in range(E * NR):
        d = next(gen)
        pos[d["user_id"], d["p_item_id"]] += 1
        neg[d["user_id"], d["n_item_id"]] += 1
    # the pointwise generators (dataset.py:18-58) on the same records
    NS, RATIO_S = 60000, 0.3
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=124)
    gen = ref_dataset._stratified_pointwise_generator(ds, RATIO_S)
    spos = np.zeros((NU, NI), np.int32); sneg = np.zeros((NU, NI), np.int32)
    for _ in range(NS):
        d = next(gen)
        (spos if d["label"] == 1.0 else sneg)[d["user_id"], d["item_id"]] += 1
    E2, RATIO_P = 60, 0.2
    nneg = int((1 - RATIO_P) / RATIO_P)
    ds = _DataStore(raw_data=raw, total_users=NU, total_items=NI, seed=125)
    gen = ref_dataset._per_pos_stratified_pointwise_generator(ds, RATIO_P)
    ppos = np.zeros((NU, NI), np.int32); pneg = np.zeros((NI, NI), np.int32)      # negatives by (record's item, negative item)
    cur = None
    for _ in range(E2 * NR * (1 + nneg)):
        d = next(gen)
        if d["label"] == 1.0:
            cur = d["item_id"]; ppos[d["user_id"], d["item_id"]] += 1
        else:
            pneg[cur, d["item_id"]] += 1
    np.savez_compressed(os.path.join(HERE, "sampler_hist.npz"), raw_user=raw["user_id"], raw_item=raw["item_id"], NU=NU, NI=NI, E=E,
                        pos=pos, neg=neg, strat_n=NS, strat_ratio=RATIO_S, strat_pos=spos, strat_neg=sneg,
                        perpos_epochs=E2, perpos_ratio=RATIO_P, perpos_pos=ppos, perpos_neg=pneg)
    print("samples", E * NR, "pos cells", int((pos > 0).sum()), "neg cells", int((neg > 0).sum()),
          "| stratified positives", int(spos.sum()), "of", NS, "| per-pos negatives", int(pneg.sum()))


if __name__ == "__main__":
    main()
