"""Shared machinery of the recommenders: a model call either runs the
forward-only kernel (no tape) or records a pending fused step (under a tape,
see openrec_amd/tf2/_lazy.py)."""
from __future__ import annotations

import os

import numpy as np

from ... import runtime as rt
from .._lazy import LazyScalar, PendingStep, active_tape

# consecutive applied steps wait in a queue of this many and run as ONE K-step device call (0: no queue)
STEP_QUEUE = int(os.environ.get("ORX_STEP_QUEUE", "32"))


class _StepQueue:
    """Snapshot of the inputs of up to STEP_QUEUE applied-but-not-yet-executed train steps of one model."""

    def __init__(self):
        self.steps, self.key, self.bufs, self.runner, self.censor = [], None, None, None, []

    def add(self, step, key, arrays, runner):
        """arrays: the per-step input arrays (all of one length B); runner(bufs, K) -> (loss[K], l2[K])"""
        cap = STEP_QUEUE
        if self.bufs is None:
            self.key, self.runner = key, runner
            self.bufs = []
            for a in arrays:
                if hasattr(a, "is_cuda") and a.is_cuda:
                    import torch
                    self.bufs.append(torch.empty((cap,) + tuple(a.shape), dtype=a.dtype, device=a.device))
                else:
                    a = np.asarray(a)
                    self.bufs.append(np.empty((cap,) + a.shape, a.dtype))
        k = len(self.steps)
        for buf, a in zip(self.bufs, arrays):
            if isinstance(buf, np.ndarray):
                buf[k] = np.asarray(a)
            else:
                buf[k].copy_(a)
        self.steps.append(step)

    def run(self):
        steps, bufs, runner, flags = self.steps, self.bufs, self.runner, self.censor
        self.steps, self.key, self.bufs, self.runner, self.censor = [], None, None, None, []
        K = len(steps)
        if K == 0:
            return
        flags = flags + [False] * (K - len(flags))
        k0 = 0
        while k0 < K:                               # consecutive steps with the same censor flag share a device call
            k1 = k0
            while k1 < K and flags[k1] == flags[k0]:
                k1 += 1
            kw = {"censor": True} if flags[k0] else {}
            loss, l2 = runner([b[k0:k1] for b in bufs], k1 - k0, **kw)
            for i in range(k0, k1):
                steps[i].values = (float(loss[i - k0]), float(l2[i - k0]))
                steps[i].trained, steps[i].queued = True, False
            k0 = k1

    def mark_censor(self, arrays):
        """UCML.censor_vec right after a queued step with the same ids: fold it into that step (ORX_CENSOR)."""
        k = len(self.steps) - 1
        if k < 0 or len(self.censor) > k:
            return False
        for buf, a in zip(self.bufs, arrays):
            if isinstance(buf, np.ndarray):
                if hasattr(a, "is_cuda") or not np.array_equal(buf[k], np.asarray(a)):
                    return False
            else:
                if not (hasattr(a, "is_cuda") and a.is_cuda and bool((buf[k] == a).all())):
                    return False
        self.censor = self.censor + [False] * (k - len(self.censor)) + [True]
        return True


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
        self._queue = _StepQueue()
        for lf in (self.user_latent_factor, self.item_latent_factor, self.item_bias):
            lf.table.pre_access = self.flush          # any host-visible access to a table first runs the queued steps

    def flush(self):
        """Run the queued train steps now (called automatically whenever their effect could be observed)."""
        if getattr(self, "_queue", None) is not None:
            self._queue.run()

    def _enqueue(self, step, key, arrays, runner):
        """Queue an applied step; returns False when queuing is off (the caller then runs it directly)."""
        if STEP_QUEUE <= 1:
            return False
        q = self._queue
        if q.steps and q.key != key:
            q.run()
        q.add(step, key, arrays, runner)
        if len(q.steps) >= STEP_QUEUE:
            q.run()
        return True

    @property
    def trainable_variables(self):
        return (self.user_latent_factor.variables + self.item_latent_factor.variables
                + self.item_bias.variables)

    @property
    def variables(self):           # (every variable of these models is trainable; a subclass that adds some is followed)
        return self.trainable_variables

    def _tables(self, flush=True):
        if flush:
            self.flush()
        return self.user_latent_factor.table, self.item_latent_factor.table, self.item_bias.table

    _score_kind = "dot"

    def evaluate(self, user_id, pos_mask, excl_mask, at=(100,)):
        """Beyond the reference API: `eval_step` of tf2_examples/bpr_citeulike.py:41-46 as one device call
        (all-item scores + AUC / NDCG / Recall; the [B, n_items] score matrix never reaches the host).  The masks as
        `Dataset.evaluation` yields them (item lists, `rt.SparseMask`) go over as lists; dense masks are accepted too
        (both as lists when they are sparse enough to be worth the host-side nonzero)."""
        U, V, b = self._tables()
        w = self.mlp.layers[0].kernel if self._score_kind == "gmf" else None
        kw = dict(kind=self._score_kind, user=U, item=V, bias=b, w=w, uid=_ids(user_id))
        if isinstance(pos_mask, rt.SparseMask) and isinstance(excl_mask, rt.SparseMask):
            return rt.rank_metrics_csr(pos_mask, excl_mask, list(at), **kw)
        return rt.rank_metrics(pos_mask, excl_mask, list(at), **kw)

    def _record(self, run_forward, run_train):
        for lf in (self.user_latent_factor, self.item_latent_factor, self.item_bias):
            lf.snapshot_pending()               # lookups made before this step see the rows as they are now (TF gathers at call time)
        step = PendingStep(self, run_forward, run_train)
        tape = active_tape()
        if tape is not None:
            tape.record(step)
        else:
            step.forward()                      # eager semantics outside a tape
        return LazyScalar(step, 0), LazyScalar(step, 1)


class PointwiseRecommender(Recommender):
    """GMF / WRMF: (user, item, label) samples; `_point_args()` gives (model name, dense kernel table or None, kwargs)."""

    def __call__(self, user_id, item_id, label):
        tape = active_tape()
        U, V, b = self._tables(flush=tape is None)
        name, w, kw = self._point_args()
        uid, iid, lab = _ids(user_id), _ids(item_id), _ids(label)
        step_holder = []

        def run_forward():
            self.flush()
            return rt.pointwise_loss(name, U, V, b, w, uid, iid, lab, **kw)

        def run_train(optimizer, no_l2):
            def runner(bufs, K):
                return rt.pointwise_step(name, optimizer, U, V, b, w, bufs[0], bufs[1], bufs[2], K=K, no_l2=no_l2, **kw)
            n = uid.numel() if hasattr(uid, "numel") else np.asarray(uid).size
            key = ("point", id(optimizer), bool(no_l2), int(n), bool(getattr(uid, "is_cuda", False)))
            if self._enqueue(step_holder[0], key, (uid, iid, np.asarray(lab, np.float32) if not hasattr(lab, "is_cuda") else lab), runner):
                return None
            loss, l2 = runner((uid, iid, lab), 1)
            return float(loss[0]), float(l2[0])

        out = self._record(run_forward, run_train)
        step_holder.append(out[0]._step)
        return out

    call = __call__


class PairwiseRecommender(Recommender):
    _model = None
    margin = 0.5

    def __call__(self, user_id, p_item_id, n_item_id):
        tape = active_tape()
        U, V, b = self._tables(flush=tape is None)        # under a tape nothing is observed yet: keep the queue
        uid, pid, nid = _ids(user_id), _ids(p_item_id), _ids(n_item_id)
        step_holder = []

        def run_forward():
            self.flush()
            return rt.pairwise_loss(self._model, U, V, b, uid, pid, nid, margin=self.margin)

        def run_train(optimizer, no_l2):
            def runner(bufs, K, censor=False):
                return rt.pairwise_step(self._model, optimizer, U, V, b, bufs[0], bufs[1], bufs[2], K=K,
                                        margin=self.margin, no_l2=no_l2, censor=censor)
            n = uid.numel() if hasattr(uid, "numel") else np.asarray(uid).size
            key = ("pair", id(optimizer), bool(no_l2), int(n), bool(getattr(uid, "is_cuda", False)), self.margin)
            if self._enqueue(step_holder[0], key, (uid, pid, nid), runner):
                return None
            loss, l2 = runner((uid, pid, nid), 1)
            return float(loss[0]), float(l2[0])

        out = self._record(run_forward, run_train)
        step_holder.append(out[0]._step)
        return out

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
