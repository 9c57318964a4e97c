"""Lazy arithmetic on looked-up rows, so that the reference's recommenders that spell their score out in raw TensorFlow ops
stay on the device path:

  ucml.py:29-40   tf.math.square(user_vec - item_vec) -> tf.math.reduce_sum(..., axis=-1, keepdims=True) -> (-d) + bias ->
                  pos - neg -> tf.maximum(margin - diff, 0) -> tf.reduce_sum          = the fused UCML step (hinge sum)
  ucml.py:50-53   -reduce_sum(square(expand_dims(user_vec, 1) - V), -1) + reshape(b)  = the all-item L2 scorer
  gmf.py:28-30    mlp(user_vec * item_vec) + item_bias -> reshape -> BinaryCrossentropy(from_logits=True)
                                                                                      = the fused GMF step
  gmf.py:36-41    squeeze(mlp(expand_dims(user_vec, 1) * V), -1) + reshape(b)         = the all-item GMF scorer

An `Expr` is a small tree over `GatheredRows` / variables / constants.  The ops that END one of these compositions
(`tf.reduce_sum`, the BCE loss object, `+ tf.reshape(bias, [-1])`) match the tree against the reference's text and record
the fused step (modules/_compose.py); a tree that matches nothing is evaluated node by node when its values are looked at (MLP /
interaction nodes on the device, element-wise glue on the fetched arrays) -- without
gradients, and says so once under a tape."""
from __future__ import annotations

import numpy as np

from .latent_factor import GatheredRows, Variable


def is_lazy(x):
    if isinstance(x, (Expr, GatheredRows, Variable)):
        return True
    if np.isscalar(x) or isinstance(x, np.ndarray):
        return False
    return getattr(x, "flat_of", None) is not None


def host(x):
    """the value of a tree node as a host array"""
    if isinstance(x, Expr):
        return x.numpy()
    if isinstance(x, Variable):
        return x.numpy()
    return np.asarray(x)


class Expr:
    __array_priority__ = 200.0

    def __init__(self, op, *args, **kw):
        self.op, self.args, self.kw = op, args, kw
        self._host = None

    # ---- arithmetic builds the tree; `+` also ends the two inference compositions
    def __add__(self, o):
        from ._compose import try_all_item_scores
        e = Expr("add", self, o)
        r = try_all_item_scores(e)
        return e if r is None else r

    def __radd__(self, o):
        from ._compose import try_all_item_scores
        e = Expr("add", o, self)
        r = try_all_item_scores(e)
        return e if r is None else r

    def __sub__(self, o): return Expr("sub", self, o)
    def __rsub__(self, o): return Expr("sub", o, self)
    def __mul__(self, o): return Expr("mul", self, o)
    def __rmul__(self, o): return Expr("mul", o, self)
    def __neg__(self): return Expr("neg", self)

    # ---- anything that looks at the values gets the host array
    def numpy(self):
        if self._host is None:
            # dlrm.py:76-100 (inference): the whole tree is one device call
            from ._compose import try_dlrm_inference
            v = try_dlrm_inference(self)
            if v is not None:
                self._host = np.asarray(v, np.float32)
                return self._host
            from .._lazy import active_tape
            if active_tape() is not None:
                # under a tape the caller expects to TRAIN through this value: a host forward has no gradients
                raise NotImplementedError(f"this expression (ending in `{self.op}`) is not one of the compositions this package runs as a fused "
                                          "device step (bpr.py / ucml.py / gmf.py / wrmf.py / dlrm.py of the reference): it would be computed on "
                                          "the host WITHOUT gradients -- evaluate it outside the GradientTape if only its values are wanted")
            op, kw = self.op, self.kw
            if op == "mlp":
                self._host = np.asarray(self.args[0].device_forward(host(self.args[1])), np.float32)          # orx_mlp_forward
                return self._host
            if op == "interact":
                self._host = np.asarray(self.args[0].device_forward([host(x) for x in self.args[1:]]), np.float32)      # orx_interact_forward
                return self._host
            if op == "concat":
                self._host = np.concatenate([np.asarray(host(x), np.float32) for x in self.args], axis=kw["axis"])
                return self._host
            a = [host(x) for x in self.args]
            if op == "add": v = a[0] + a[1]
            elif op == "sub": v = a[0] - a[1]
            elif op == "mul": v = a[0] * a[1]
            elif op == "neg": v = -a[0]
            elif op == "square": v = np.square(a[0])
            elif op == "reduce_sum": v = np.sum(a[0], axis=kw.get("axis"), keepdims=kw.get("keepdims", False), dtype=np.float32)
            elif op == "maximum": v = np.maximum(a[0], a[1])
            elif op == "expand_dims": v = np.expand_dims(a[0], kw["axis"])
            elif op == "squeeze": v = np.squeeze(a[0], axis=kw.get("axis"))
            elif op == "reshape": v = np.reshape(a[0], kw["shape"])
            elif op == "dense1": v = a[1] @ self.args[0].layers[0].kernel.read()
            elif op == "clip": v = np.clip(a[0], kw["lo"], kw["hi"])
            else: raise NotImplementedError(op)
            self._host = np.asarray(v, np.float32)
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __float__(self):
        return float(self.numpy())

    @property
    def shape(self):
        return self.numpy().shape

    def __getitem__(self, k):
        return self.numpy()[k]

    def __repr__(self):
        return f"<Expr {self.op}>"


# ------------------------------------------------------------------ the TensorFlow ops that build / end a tree
def square(x):
    return Expr("square", x) if is_lazy(x) else np.square(np.asarray(x))


def reduce_sum(x, axis=None, keepdims=False):
    if not is_lazy(x):
        return np.sum(np.asarray(x), axis=axis, keepdims=keepdims)
    e = Expr("reduce_sum", x, axis=axis, keepdims=keepdims)
    if axis is None:
        from ._compose import try_ucml_loss
        r = try_ucml_loss(e)
        if r is not None:
            return r
    return e


def maximum(a, b):
    return Expr("maximum", a, b) if (is_lazy(a) or is_lazy(b)) else np.maximum(np.asarray(a), np.asarray(b))


def concat(values, axis):
    values = list(values)
    return Expr("concat", *values, axis=axis) if any(is_lazy(v) for v in values) else np.concatenate([np.asarray(v) for v in values], axis=axis)


def clip_by_value(x, lo, hi):
    return Expr("clip", x, lo=float(lo), hi=float(hi)) if is_lazy(x) else np.clip(np.asarray(x), lo, hi)


def unstack(x, axis=0):
    a = np.asarray(x.numpy() if hasattr(x, "numpy") and not isinstance(x, np.ndarray) else x)
    return [np.take(a, k, axis=axis) for k in range(a.shape[axis])]


def expand_dims(x, axis):
    return Expr("expand_dims", x, axis=axis) if is_lazy(x) else np.expand_dims(np.asarray(x), axis)


def squeeze(x, axis=None):
    return Expr("squeeze", x, axis=axis) if is_lazy(x) else np.squeeze(np.asarray(x), axis=axis)


class BinaryCrossentropy:
    """tf.keras.losses.BinaryCrossentropy as gmf.py:20 makes it (from_logits=True, mean over the batch)"""

    def __init__(self, from_logits=False, **_):
        self.from_logits = from_logits

    def __call__(self, y_true, y_pred, sample_weight=None):
        if isinstance(y_pred, Expr) and self.from_logits and sample_weight is None:
            from ._compose import try_gmf_loss
            r = try_gmf_loss(y_true, y_pred)
            if r is not None:
                return r
        if isinstance(y_pred, Expr) and not self.from_logits and sample_weight is None:
            from ._compose import try_dlrm_loss          # dlrm.py:54-55, :72-73
            r = try_dlrm_loss("bce", y_true, y_pred)
            if r is not None:
                return r
        y, x = np.asarray(y_true, np.float32).reshape(-1), host(y_pred).reshape(-1).astype(np.float32)
        if self.from_logits:
            return np.float32(np.mean(np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))))
        eps = 1e-7
        p = np.clip(x, eps, 1 - eps)
        return np.float32(np.mean(-(y * np.log(p + eps) + (1 - y) * np.log(1 - p + eps))))


class MeanSquaredError:
    """tf.keras.losses.MeanSquaredError as dlrm.py:52-53 makes it (mean over the batch)"""

    def __call__(self, y_true, y_pred, sample_weight=None):
        if isinstance(y_pred, Expr) and sample_weight is None:
            from ._compose import try_dlrm_loss          # dlrm.py:72-73
            r = try_dlrm_loss("mse", y_true, y_pred)
            if r is not None:
                return r
        y, x = np.asarray(y_true, np.float32).reshape(-1), host(y_pred).reshape(-1).astype(np.float32)
        return np.float32(np.mean((y - x) ** 2))
