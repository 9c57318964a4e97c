"""`openrec.tf2.metrics` surface (ranking_metrics.py:8-69, dict_mean.py:4-32): per-user AUC /
NDCG@k / Recall@k from full score rows and masks, computed on the device."""
from __future__ import annotations

import numpy as np

from ... import runtime as rt


def _host(x, dtype):
    if hasattr(x, "numpy") and not isinstance(x, np.ndarray):
        x = x.numpy()
    return np.ascontiguousarray(x, dtype=dtype)


def _rank(pos_mask, pred, excl_mask, at):
    """Item-list masks (what `Dataset.evaluation` yields) go to the device as lists, and scores that `inference` left in
    device memory are read there; dense masks / host scores take the byte-mask entry point."""
    if isinstance(pos_mask, rt.SparseMask) and isinstance(excl_mask, rt.SparseMask):
        return rt.rank_metrics_csr(pos_mask, excl_mask, at, pred=pred if isinstance(pred, rt.DeviceScores) else _host(pred, np.float32))
    return rt.rank_metrics(_host(pos_mask, np.uint8), _host(excl_mask, np.uint8), at, pred=_host(pred, np.float32))


def AUC(pos_mask, pred, excl_mask):
    return _rank(pos_mask, pred, excl_mask, [1.0])["auc"]


def NDCG(pos_mask, pred, excl_mask, at=[100]):
    return _rank(pos_mask, pred, excl_mask, at)["ndcg"]


def Recall(pos_mask, pred, excl_mask, at=[100]):
    return _rank(pos_mask, pred, excl_mask, at)["recall"]


class DictMean:
    """Running mean over the leading (batch) axis of every entry (dict_mean.py:4-32)."""

    def __init__(self, state_shape):
        self._shapes = dict(state_shape)
        self.reset_states()

    def reset_states(self):
        self._sum = {k: np.zeros(tuple(s), np.float32) for k, s in self._shapes.items()}
        self._count = {k: 0.0 for k in self._shapes}

    def update_state(self, state):
        for k, v in state.items():
            v = np.asarray(v.numpy() if hasattr(v, "numpy") else v, np.float32)
            self._sum[k] = self._sum[k] + v.sum(axis=0)
            self._count[k] += float(v.shape[0])

    def result(self):
        from ..compat import HostTensor          # (`.numpy()` on every entry, as bpr_citeulike.py:64-65 prints them)
        return {k: HostTensor(self._sum[k] / np.float32(self._count[k])) for k in self._sum}
