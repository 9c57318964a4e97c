"""PairwiseLogLoss on already gathered vectors
(openrec/tf2/modules/pairwise_log_loss.py:4-34).  The recommenders do not go
through this module on the training path (gather, score, loss and update are
one fused kernel); it exists for API parity and computes on the host arrays it
is given."""
import numpy as np


class PairwiseLogLoss:

    def __call__(self, user_vec, p_item_vec, n_item_vec, p_item_bias=None, n_item_bias=None):
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
