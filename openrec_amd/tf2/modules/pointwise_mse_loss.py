"""PointwiseMSELoss on already gathered vectors
(openrec/tf2/modules/pointwise_mse_loss.py:4-31).  Given the three lookups of wrmf.py:23-25 it records the fused
WRMF step (`WRMF.__call__` of this package; `sigmoid=True` is a flag of the same fused kernel); plain arrays compute on the host."""
import numpy as np

from . import _compose


class PointwiseMSELoss:

    def __init__(self, a=1.0, b=1.0, sigmoid=False):
        self._a, self._b, self._sigmoid = a, b, sigmoid

    def __call__(self, user_vec, item_vec, item_bias, label):
        fused = _compose.pointwise_step_of(user_vec, item_vec, item_bias, label, self._a, self._b, sigmoid=bool(self._sigmoid))
        if fused is not None:
            return fused[0]
        if any(isinstance(x, _compose.GatheredRows) for x in (user_vec, item_vec)):
            _compose.host_fallback("PointwiseMSELoss")
        u, i = np.asarray(user_vec, np.float32), np.asarray(item_vec, np.float32)
        label = np.asarray(label, np.float32).reshape(-1)
        pred = (u * i).sum(axis=1) + np.asarray(item_bias, np.float32).reshape(-1)
        if self._sigmoid:
            pred = 1.0 / (1.0 + np.exp(-pred))
        weight = np.float32(self._a - self._b) * label + np.float32(self._b)
        return np.float32((weight * np.square(label - pred)).sum())

    def call(self, inputs):
        return self.__call__(*inputs)
