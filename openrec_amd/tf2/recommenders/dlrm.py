"""DLRM (openrec/tf2/recommenders/dlrm.py:6-100): same constructor arguments,
`__call__(dense_features, sparse_features, label) -> loss` and
`inference(dense_features, sparse_features) -> [B]`.

`reference_compat` (an extra keyword, default True) keeps the reference's
feature-interaction behaviour bit for bit -- including the triangle bug that
makes the interaction output zero; set it to False for the
evidently intended pairwise dot products."""
from __future__ import annotations

import sys

import numpy as np

from ... import runtime as rt
from .._lazy import LazyScalar, PendingStep, active_tape
from ..modules.latent_factor import Variable
from ._base import _StepQueue, STEP_QUEUE


class DLRM:

    def __init__(self, m_spa, ln_emb, ln_bot, ln_top, arch_interaction_op='dot', arch_interaction_itself=False,
                 sigmoid_bot=False, sigmoid_top=True, loss_func='mse', loss_threshold=0.0,
                 reference_compat=True, dense_dim=13, ctx=None, seed=0):
        if arch_interaction_op != 'dot':
            # dlrm.py:45 reads self._arch_interaction_op, which is never assigned: every value other
            # than 'dot' raises AttributeError in the reference
            raise AttributeError("'DLRM' object has no attribute '_arch_interaction_op'")
        if loss_func not in ('mse', 'bce'):
            sys.exit("ERROR: loss_func=" + loss_func + " is not supported")       # dlrm.py:56-61
        self._model = rt.DLRMModel(m_spa, ln_emb, ln_bot, ln_top, dense_dim, arch_interaction_itself, sigmoid_bot,
                                   sigmoid_top, loss_func, loss_threshold, reference_compat, seed, ctx)
        self._queue = _StepQueue()          # applied train steps wait here and run as one K-step device call
        self._hooked = set()

    def flush(self):
        self._queue.run()

    def _param(self, kind, layer=0):
        """parameter table whose host-visible accesses first run the queued steps"""
        t = self._model.param(kind, layer)
        t.pre_access = self.flush
        return t

    @property
    def trainable_variables(self):
        if getattr(self, "_vars", None) is not None:        # the loop asks for this list twice per step
            return self._vars
        m = self._model
        out = [Variable(self._param("emb"), "latent_factors/embeddings")]
        for nm, n in (("bot", len(m.ln_bot)), ("top", len(m.ln_top))):
            for l in range(n):
                out.append(Variable(self._param(nm + "_w", l), f"mlp_{nm}/dense_{l}/kernel"))
                out.append(Variable(self._param(nm + "_b", l), f"mlp_{nm}/dense_{l}/bias"))
        self._vars = out
        return out

    def __call__(self, dense_features, sparse_features, label):
        m = self._model
        d, s, y = (np.asarray(x.numpy() if hasattr(x, "numpy") else x) for x in (dense_features, sparse_features, label))

        def run_forward():
            self.flush()
            p = m.inference(d, s).astype(np.float32)           # (already clipped to the loss threshold, dlrm.py:97-98)
            yy = y.astype(np.float32).reshape(-1)
            if m.loss_func == "bce":                            # keras BinaryCrossentropy on probabilities (dlrm.py:54-55, :72-73)
                pc = np.clip(p, np.float32(1e-7), np.float32(1.0 - 1e-7))
                return (float(np.mean(-(yy * np.log(pc) + (1.0 - yy) * np.log(1.0 - pc)))), 0.0)
            return (float(np.mean((yy - p) ** 2)), 0.0)        # keras MeanSquaredError (dlrm.py:52-53)

        def run_train(optimizer, no_l2):
            def runner(bufs, K):
                loss = m.step(optimizer, bufs[0].reshape(-1, d.shape[-1]), bufs[1].reshape(-1, s.shape[-1]), bufs[2].reshape(-1), K=K)
                return loss, np.zeros(K, np.float32)
            if STEP_QUEUE > 1:
                q = self._queue
                key = ("dlrm", id(optimizer), d.shape, s.shape)
                if q.steps and q.key != key:
                    q.run()
                q.add(step, key, (np.asarray(d, np.float32), np.asarray(s, np.int32), np.asarray(y, np.float32).reshape(-1)), runner)
                if len(q.steps) >= STEP_QUEUE:
                    q.run()
                return None
            loss = m.step(optimizer, d, s, y, K=1)
            return float(loss[0]), 0.0

        step = PendingStep(self, run_forward, run_train)
        step.user_latent_factor = None
        tape = active_tape()
        if tape is not None:
            tape.record(step)
        return LazyScalar(step, 0)

    call = __call__

    # `optimizer.apply_gradients` looks the context up through this attribute
    @property
    def user_latent_factor(self):
        class _T:
            table = self._model.param("emb")       # context lookup only: no host-visible access, no flush
        return _T

    def inference(self, dense_features, sparse_features):
        self.flush()
        return self._model.inference(dense_features, sparse_features)
