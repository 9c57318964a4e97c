"""Deferred execution glue: lets the reference's tape-style train step

    with tf.GradientTape() as tape:
        loss_value = model(user_id, p_item_id, n_item_id)          # (loss, l2_loss)
    gradients = tape.gradient(loss_value, model.trainable_variables)
    optimizer.apply_gradients(zip(gradients, model.trainable_variables))

(tf2_examples/bpr_citeulike.py:33-39) run as ONE fused device call: the model
call inside a tape records a pending step and returns lazy scalars, the tape
hands out gradient tokens, and `apply_gradients` launches the fused
forward+backward+update.  Reading a lazy scalar before that (no optimizer step)
falls back to the forward-only kernel.

Consecutive train steps are additionally QUEUED (their inputs do not depend on
each other's results): `apply_gradients` snapshots the ids and returns, and the
queue runs as one K-step device call when it is full or when anything observes
the model (a loss value, a table, inference, a different optimizer ...).  A
per-step call costs ~220 us of launch / synchronization latency at B = 65536
against ~35 us per step inside a K-step call, so this is what makes the
reference's one-batch-at-a-time training loop fast without changing it."""
from __future__ import annotations

import numpy as np

_tape_stack = []


def active_tape():
    return _tape_stack[-1] if _tape_stack else None


class PendingStep:
    """One recorded model call: enough to run either the forward-only kernel or
    the fused train step."""

    def __init__(self, model, run_forward, run_train):
        self.model = model
        self._run_forward, self._run_train = run_forward, run_train
        self.values = None            # (loss, l2_loss) once known
        self.trained = False
        self.queued = False           # applied, but still waiting in the model's step queue
        self.objective = "sum"        # which parts of the tuple the tape differentiated

    def forward(self):
        if self.values is None and self.queued:
            self.model.flush()        # the queued K-step call fills in `values`
        if self.values is None:
            self.values = self._run_forward()
        return self.values

    def train(self, optimizer, no_l2):
        if self.trained or self.queued:
            raise RuntimeError("this recorded step was already applied")
        out = self._run_train(optimizer, no_l2)
        if out is None:               # queued: values arrive with the flush
            self.queued = True
            return None
        self.values = out
        self.trained = True
        return self.values


class LazyScalar:
    """A scalar that materializes on demand (`.numpy()`, float(), arithmetic)."""

    def __init__(self, step, index):
        self._step, self._index = step, index

    def numpy(self):
        return np.float32(self._step.forward()[self._index])

    def __float__(self):
        return float(self.numpy())

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.numpy())
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f"<LazyScalar {float(self):.6g}>"

    def __add__(self, o): return float(self) + float(o)
    __radd__ = __add__
    def __mul__(self, o): return float(self) * float(o)
    __rmul__ = __mul__
    def __sub__(self, o): return float(self) - float(o)
    def __truediv__(self, o): return float(self) / float(o)
    def __format__(self, spec): return format(float(self), spec)


class GradToken:
    """Stands for d(objective)/d(variable) of a recorded step; consumed by
    `optimizer.apply_gradients`."""

    def __init__(self, step, variable, no_l2):
        self.step, self.variable, self.no_l2 = step, variable, no_l2


class GradientTape:
    """`tf.GradientTape` stand-in (tf2_examples/bpr_citeulike.py:34)."""

    def __init__(self, persistent=False, watch_accessed_variables=True):
        self.steps = []

    def __enter__(self):
        _tape_stack.append(self)
        return self

    def __exit__(self, *exc):
        _tape_stack.pop()
        return False

    def record(self, step):
        self.steps.append(step)

    def gradient(self, target, sources, **_):
        """`target` is what the model call returned: the (loss, l2_loss) tuple
        (TF sums a nested target) or only its loss element."""
        flat = list(target) if isinstance(target, (tuple, list)) else [target]
        from .modules._compose import L2Sum
        for i, t in enumerate(flat):            # tf.nn.l2_loss of a composition's lookups = the step's second output
            if isinstance(t, L2Sum):
                st = t.resolve()
                if st is None or st not in self.steps:
                    if getattr(t, "scale", 1.0) != 1.0:
                        raise NotImplementedError(f"tape.gradient: the l2 term is scaled by {t.scale:g}; the fused step takes tf.nn.l2_loss of its "
                                                  "lookups with weight 1 (the reference's objective, bpr.py:35-37) or not at all (differentiate `loss` alone)")
                    raise NotImplementedError("tape.gradient: this l2 term is not the l2_loss of one recorded step's lookups "
                                              "(bpr.py:35 / wrmf.py:32 sum tf.nn.l2_loss over exactly the looked-up vectors)")
                flat[i] = LazyScalar(st, 1)
        lazies = [t for t in flat if isinstance(t, LazyScalar)]
        if not lazies:
            raise ValueError("tape.gradient: target does not come from a recommender call under this tape")
        step = lazies[0]._step
        idx = sorted({t._index for t in lazies if t._step is step})
        if idx == [0, 1]:
            no_l2 = False
        elif idx == [0]:
            no_l2 = True
        else:
            raise NotImplementedError("only d(loss + l2_loss) and d(loss) are fused")
        return [GradToken(step, v, no_l2) for v in sources]
