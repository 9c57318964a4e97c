"""PointwiseMSELoss on already gathered vectors
(openrec/tf2/modules/pointwise_mse_loss.py:4-31); API parity, host arrays."""
import numpy as np


class PointwiseMSELoss:

    def __init__(self, a=1.0, b=1.0, sigmoid=False):
        self._a, self._b, self._sigmoid = a, b, sigmoid

    def __call__(self, user_vec, item_vec, item_bias, label):
        u, i = np.asarray(user_vec, np.float32), np.asarray(item_vec, np.float32)
        label = np.asarray(label, np.float32).reshape(-1)
        pred = (u * i).sum(axis=1) + np.asarray(item_bias, np.float32).reshape(-1)
        if self._sigmoid:
            pred = 1.0 / (1.0 + np.exp(-pred))
        weight = np.float32(self._a - self._b) * label + np.float32(self._b)
        return np.float32((weight * np.square(label - pred)).sum())
