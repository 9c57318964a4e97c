"""The handful of TensorFlow symbols that `tf2_examples/*.py` touch around the
hot path, so that a script shaped like
tf2_examples/bpr_citeulike.py runs with only its imports changed:

    from openrec_amd.tf2.compat import tf, optimizers
    from openrec_amd.tf2.recommenders import BPR

`install()` additionally registers this module as `tensorflow` in sys.modules --
only when no real TensorFlow is importable; it never shadows one."""
from __future__ import annotations

import sys
import types
import weakref

import numpy as np

from .. import runtime as rt
from ._lazy import GradientTape, GradToken, LazyScalar


def function(fn=None, **_):
    """tf.function: the fused step is already one device call, nothing to trace."""
    if fn is None:
        return lambda f: f
    return fn


def constant(value, dtype=None):
    return np.asarray(value, dtype=dtype)


int32, float32, bool_ = np.int32, np.float32, np.bool_


class _Optimizer:
    """Keras optimizer facade; the device-side state lives in runtime.Optimizer."""
    _kind = None

    def __init__(self):
        self._native = None
        self._models = weakref.WeakSet()       # models with steps applied through this optimizer (their queues hold it)
        self._lr = None
        self._iterations = 0

    def _make(self, ctx):
        raise NotImplementedError

    def native(self, ctx=None):
        if self._native is None:
            self._native = self._make(ctx)
        return self._native

    def _flush_models(self):
        for m in list(self._models):
            flush = getattr(m, "flush", None)
            if flush is not None:
                flush()

    # Keras `optimizer.learning_rate` (also `.lr`): assignable at any time.  Steps already applied -- queued or not --
    # keep the rate they were applied with: the queues run first, then the device-side optimizer takes the new rate.
    @property
    def learning_rate(self):
        return self._lr

    @learning_rate.setter
    def learning_rate(self, value):
        value = float(value)
        if self._native is not None and value != self._lr:
            self._flush_models()
            self._native.set_lr(value)
        self._lr = value

    lr = learning_rate

    @property
    def iterations(self):
        """Keras `optimizer.iterations`: steps applied so far (the queued ones included).  An Adam optimizer keeps the
        counter natively (it is part of a checkpoint, runtime.load_checkpoint restores it): the queues run, then the
        native counter answers."""
        if self._native is not None and self._native.kind == "adam":
            self._flush_models()
            return self._native.step
        return self._iterations

    def apply_gradients(self, grads_and_vars):
        """Consumes the tokens of `GradientTape.gradient`: all tokens of one recorded
        step trigger ONE fused forward+backward+update call."""
        steps = []
        for g, v in grads_and_vars:
            if not isinstance(g, GradToken):
                raise TypeError("apply_gradients expects gradients produced by openrec_amd's GradientTape")
            if g.step not in [s for s, _ in steps]:
                steps.append((g.step, g.no_l2))
        for step, no_l2 in steps:
            ctx = step.model.user_latent_factor.table.ctx
            self._models.add(step.model)
            step.train(self.native(ctx), no_l2)
            self._iterations += 1


class SGD(_Optimizer):
    def __init__(self, learning_rate=0.01, momentum=0.0, **_):
        super().__init__()
        if momentum:
            raise NotImplementedError("momentum SGD is not on the reference's hot path")
        self.learning_rate = learning_rate

    def _make(self, ctx):
        return rt.Optimizer.sgd(self.learning_rate, ctx=ctx)


class Adagrad(_Optimizer):
    def __init__(self, learning_rate=0.001, initial_accumulator_value=0.1, epsilon=1e-7, **_):
        super().__init__()
        self.learning_rate, self.iav, self.epsilon = learning_rate, initial_accumulator_value, epsilon

    def _make(self, ctx):
        return rt.Optimizer.adagrad(self.learning_rate, self.iav, self.epsilon, ctx=ctx)


class Adam(_Optimizer):
    """keras.optimizers.Adam() as used by tf2_examples/bpr_citeulike.py:31 (TF-2.0
    dense-decay sparse apply)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **_):
        super().__init__()
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon

    def _make(self, ctx):
        return rt.Optimizer.adam(self.learning_rate, self.beta_1, self.beta_2, self.epsilon, ctx=ctx)


class Mean:
    """tf.keras.metrics.Mean; `update_state((loss, l2))` averages the two scalars
    together, like the reference script does (bpr_citeulike.py:54)."""

    def __init__(self):
        self.reset_states()

    def update_state(self, values):
        for v in (values if isinstance(values, (tuple, list)) else [values]):
            if isinstance(v, LazyScalar):
                self._pending.append(v)         # reading it now would force the queued train steps to run
                if len(self._pending) > 4096:
                    self._drain()
                continue
            a = np.asarray(v.numpy() if hasattr(v, "numpy") else v, np.float64)
            self._sum += float(a.sum())
            self._n += a.size

    __call__ = update_state

    def _drain(self):
        for v in self._pending:
            self._sum += float(v.numpy())
            self._n += 1
        self._pending = []

    def result(self):
        self._drain()
        return np.float32(self._sum / max(self._n, 1))

    def reset_states(self):
        self._sum, self._n, self._pending = 0.0, 0, []


optimizers = types.SimpleNamespace(SGD=SGD, Adagrad=Adagrad, Adam=Adam)
keras = types.SimpleNamespace(optimizers=optimizers, metrics=types.SimpleNamespace(Mean=Mean))
tf = types.SimpleNamespace(function=function, GradientTape=GradientTape, constant=constant, keras=keras,
                           int32=int32, float32=float32, bool=bool_)


def install():
    """Make `import tensorflow as tf` / `from tensorflow.keras import optimizers`
    resolve to this shim when (and only when) TensorFlow is absent."""
    try:
        import tensorflow  # noqa: F401
        return False
    except Exception:
        pass
    mod = types.ModuleType("tensorflow")
    for k, v in vars(tf).items():
        setattr(mod, k, v)
    kmod = types.ModuleType("tensorflow.keras")
    kmod.optimizers, kmod.metrics = optimizers, keras.metrics
    omod = types.ModuleType("tensorflow.keras.optimizers")
    omod.SGD, omod.Adagrad, omod.Adam = SGD, Adagrad, Adam
    mod.keras = kmod
    sys.modules["tensorflow"] = mod
    sys.modules["tensorflow.keras"] = kmod
    sys.modules["tensorflow.keras.optimizers"] = omod
    return True
