"""CPU oracle of the reference's ranking metrics (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see
numpy_oracle.py).  Restates openrec/tf2/metrics/ranking_metrics.py:8-69 user by user in fp32."""
import numpy as np


def _rank_above(pred, pos_mask, excl_mask):
    v = np.exp(pred.astype(np.float32)) * (~excl_mask).astype(np.float32)       # :33, :56
    pos = v[pos_mask]
    return (v[None, :] > pos[:, None]).sum(1).astype(np.float32)                 # :35, :58


def auc(pos_mask, pred, excl_mask):
    out = []
    for pm, pr, em in zip(pos_mask.astype(bool), pred.astype(np.float32), excl_mask.astype(bool)):
        ev = ~(pm | em)                                                          # :14
        cnt = (pr[ev][None, :] <= pr[pm][:, None]).sum()                         # :18
        with np.errstate(invalid="ignore", divide="ignore"):
            out.append(np.float32(cnt) / np.float32(pm.sum() * ev.sum()))        # :19
    return np.array(out, np.float32)


def ndcg(pos_mask, pred, excl_mask, at=(100,)):
    out = []
    for pm, pr, em in zip(pos_mask.astype(bool), pred, excl_mask.astype(bool)):
        ra = _rank_above(pr, pm, em)
        lr = np.float32(1) / (np.log(ra + np.float32(2)) / np.log(np.float32(2)))   # :38
        out.append([np.float32((lr * (ra < a)).sum()) for a in at])             # :40
    return np.array(out, np.float32)


def recall(pos_mask, pred, excl_mask, at=(100,)):
    out = []
    for pm, pr, em in zip(pos_mask.astype(bool), pred, excl_mask.astype(bool)):
        ra = _rank_above(pr, pm, em)
        with np.errstate(invalid="ignore", divide="ignore"):
            out.append([np.float32((ra < a).sum()) / np.float32(pm.sum()) for a in at])   # :62-63
    return np.array(out, np.float32)
