"""PairwiseLogLoss on already gathered vectors
(openrec/tf2/modules/pairwise_log_loss.py:4-34).  Given the five lookups of
bpr.py:23-27 (`LatentFactor.__call__` results: user, positive / negative item rows and
their biases) it records the SAME fused step `BPR.__call__` records -- gather, score,
loss, gradients and update in one kernel once a tape and an optimizer ask for them.
Given anything else (plain arrays, no biases) it computes the loss on the host."""
import numpy as np

from . import _compose


class PairwiseLogLoss:

    def __call__(self, user_vec, p_item_vec, n_item_vec, p_item_bias=None, n_item_bias=None):
        fused = _compose.pairwise_step_of(user_vec, p_item_vec, n_item_vec, p_item_bias, n_item_bias)
        if fused is not None:
            return fused[0]
        if any(isinstance(x, _compose.GatheredRows) for x in (user_vec, p_item_vec, n_item_vec)):
            _compose.host_fallback("PairwiseLogLoss")
        u, p, n = (np.asarray(x, np.float32) for x in (user_vec, p_item_vec, n_item_vec))
        dot_user_pos = (u * p).sum(axis=1, keepdims=True)
        dot_user_neg = (u * n).sum(axis=1, keepdims=True)
        if p_item_bias is not None:
            dot_user_pos = dot_user_pos + np.asarray(p_item_bias, np.float32).reshape(-1, 1)
        if n_item_bias is not None:
            dot_user_neg = dot_user_neg + np.asarray(n_item_bias, np.float32).reshape(-1, 1)
        x = np.maximum(dot_user_pos - dot_user_neg, np.float32(-30.0))
        log_sig = -(np.maximum(-x, 0) + np.log1p(np.exp(-np.abs(x))))
        return np.float32(-log_sig.mean())

    def call(self, inputs):
        return self.__call__(*inputs)
