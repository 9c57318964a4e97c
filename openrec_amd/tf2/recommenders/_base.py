"""Shared machinery of the recommenders: a model call either runs the
forward-only kernel (no tape) or records a pending fused step (under a tape,
see openrec_amd/tf2/_lazy.py)."""
from __future__ import annotations

import numpy as np

from ... import runtime as rt
from .._lazy import LazyScalar, PendingStep, active_tape


def _ids(x):
    if hasattr(x, "numpy") and not isinstance(x, np.ndarray) and not getattr(x, "is_cuda", False):
        x = x.numpy()
    return x


class Recommender:
    """Base of BPR / UCML / GMF / WRMF: three tables exactly as in the reference
    constructors (user_latent_factor, item_latent_factor, item_bias)."""

    def _build_tables(self, dim_user_embed, dim_item_embed, total_users, total_items, ctx=None):
        from ..modules import LatentFactor
        if dim_user_embed != dim_item_embed:
            # the reference multiplies / subtracts the two vectors element-wise, so unequal
            # dims fail inside TF at the first call; fail at construction instead
            raise ValueError("dim_user_embed and dim_item_embed must be equal")
        self.user_latent_factor = LatentFactor(num_instances=total_users, dim=dim_user_embed,
                                               name='user_latent_factor', ctx=ctx)
        self.item_latent_factor = LatentFactor(num_instances=total_items, dim=dim_item_embed,
                                               name='item_latent_factor', ctx=ctx)
        self.item_bias = LatentFactor(num_instances=total_items, dim=1, name='item_bias', ctx=ctx)

    @property
    def trainable_variables(self):
        return (self.user_latent_factor.variables + self.item_latent_factor.variables
                + self.item_bias.variables)

    variables = trainable_variables

    def _tables(self):
        return self.user_latent_factor.table, self.item_latent_factor.table, self.item_bias.table

    _score_kind = "dot"

    def evaluate(self, user_id, pos_mask, excl_mask, at=(100,)):
        """Beyond the reference API: `eval_step` of tf2_examples/bpr_citeulike.py:41-46 as one device call
        (all-item scores + AUC / NDCG / Recall; the [B, n_items] score matrix never reaches the host)."""
        U, V, b = self._tables()
        w = self.mlp.layers[0].kernel if self._score_kind == "gmf" else None
        return rt.rank_metrics(pos_mask, excl_mask, list(at), kind=self._score_kind, user=U, item=V, bias=b, w=w,
                               uid=_ids(user_id))

    def _record(self, run_forward, run_train):
        step = PendingStep(self, run_forward, run_train)
        tape = active_tape()
        if tape is not None:
            tape.record(step)
        else:
            step.forward()                      # eager semantics outside a tape
        return LazyScalar(step, 0), LazyScalar(step, 1)


class PairwiseRecommender(Recommender):
    _model = None
    margin = 0.5

    def __call__(self, user_id, p_item_id, n_item_id):
        U, V, b = self._tables()
        uid, pid, nid = _ids(user_id), _ids(p_item_id), _ids(n_item_id)

        def run_forward():
            return rt.pairwise_loss(self._model, U, V, b, uid, pid, nid, margin=self.margin)

        def run_train(optimizer, no_l2):
            loss, l2 = rt.pairwise_step(self._model, optimizer, U, V, b, uid, pid, nid, K=1,
                                        margin=self.margin, no_l2=no_l2)
            return float(loss[0]), float(l2[0])

        return self._record(run_forward, run_train)

    call = __call__

    def train_steps(self, optimizer, user_id, p_item_id, n_item_id, K=None, want_loss=True, censor=False):
        """Beyond the reference API: K consecutive fused steps in one device call
        (ids shaped [K, B]); the path `bench.py` measures."""
        U, V, b = self._tables()
        uid, pid, nid = _ids(user_id), _ids(p_item_id), _ids(n_item_id)
        if K is None:
            K = uid.shape[0] if getattr(uid, "ndim", 1) == 2 else 1
        opt = optimizer.native(U.ctx) if hasattr(optimizer, "native") else optimizer
        return rt.pairwise_step(self._model, opt, U, V, b, uid, pid, nid, K=K, margin=self.margin,
                                want_loss=want_loss, censor=censor)
