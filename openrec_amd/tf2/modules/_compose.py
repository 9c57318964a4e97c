"""Module-level compositions that run as ONE fused device step.

The reference builds its recommenders out of modules (recommenders/bpr.py:21-37, wrmf.py:21-34): `LatentFactor` lookups,
a loss module on the looked-up vectors, `tf.nn.l2_loss` of the same vectors, then `tape.gradient((loss, l2_loss), vars)` and
`optimizer.apply_gradients` (tf2_examples/bpr_citeulike.py:33-39).  A user who writes such a model by hand gets the same
fused kernel as `openrec_amd.tf2.recommenders.BPR / WRMF`: the loss modules recognise lookups of three (two) tables and
record a pending step on a model object made of those very `LatentFactor`s; the l2 terms resolve to the step's second
output.  What is NOT a recognised composition computes on the host, without gradients -- and says so once."""
from __future__ import annotations

import warnings
import weakref

import numpy as np

from .latent_factor import GatheredRows

_models = weakref.WeakValueDictionary()     # introspection only: the strong reference lives on the user LatentFactor
_warned = set()


def host_fallback(what):
    if what not in _warned:
        _warned.add(what)
        warnings.warn(f"{what}: this call is not one of the compositions this package runs as a fused device step -- computed "
                      "on the host, WITHOUT gradients (a tape over it cannot train)", RuntimeWarning, stacklevel=3)


def _same_ids(a, b):
    if a is b:
        return True
    if getattr(a, "is_cuda", False) or getattr(b, "is_cuda", False):
        return a is b
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool((a == b).all())


def _chain(table, flush):
    prev = table.pre_access
    if prev is None:
        table.pre_access = flush
    elif prev is not flush:
        def both(prev=prev, flush=flush):
            prev(); flush()
        table.pre_access = both


def composed_model(kind, user_lf, item_lf, bias_lf, **kw):
    """the model object (step queue, tables) behind a hand-made composition of these three LatentFactors"""
    from ..recommenders._base import PairwiseRecommender, PointwiseRecommender, _StepQueue
    # the composed model lives exactly as long as its user LatentFactor (it holds the other two alive, so their ids stay
    # unique while it exists): a dict on the module would keep every composition's tables in HBM for the life of the process
    key = (kind, id(user_lf), id(item_lf), id(bias_lf), tuple(sorted(kw.items())))
    owned = user_lf.__dict__.setdefault("_composed", {})
    m = owned.get(key)
    if m is None:
        if kind == "bpr":
            class _Composed(PairwiseRecommender):
                _model = "bpr"
        else:
            class _Composed(PointwiseRecommender):
                def _point_args(self):
                    return "wrmf", None, dict(a=kw["a"], b_w=kw["b"])
        m = _Composed.__new__(_Composed)
        m.user_latent_factor, m.item_latent_factor, m.item_bias = user_lf, item_lf, bias_lf
        m._queue = _StepQueue()
        wm = weakref.ref(m)                    # (an optimizer keeps the TABLES alive; their hook must not keep the model)

        def flush():
            mm = wm()
            if mm is not None:
                mm.flush()
        for lf in (user_lf, item_lf, bias_lf):
            _chain(lf.table, flush)
        owned[key] = m
        _models[key] = m
    return m


def _tag(step, rows):
    step.l2_rows = rows
    for r in rows:
        r._l2_step = step


def pairwise_step_of(user_vec, p_item_vec, n_item_vec, p_item_bias, n_item_bias):
    """-> (loss, l2) lazy scalars of the fused BPR step these five lookups describe, or None"""
    rows = (user_vec, p_item_vec, n_item_vec, p_item_bias, n_item_bias)
    if not all(isinstance(r, GatheredRows) for r in rows):
        return None
    U, V, b = user_vec.factor, p_item_vec.factor, p_item_bias.factor
    if n_item_vec.factor is not V or n_item_bias.factor is not b or b.dim != 1 or U.dim != V.dim or b.num_instances != V.num_instances:
        return None
    if not (_same_ids(p_item_vec.ids, p_item_bias.ids) and _same_ids(n_item_vec.ids, n_item_bias.ids)):
        return None
    if len(np.shape(user_vec.ids) if not hasattr(user_vec.ids, "shape") else user_vec.ids.shape) != 1:
        return None
    out = composed_model("bpr", U, V, b)(user_vec.ids, p_item_vec.ids, n_item_vec.ids)
    _tag(out[0]._step, (user_vec, p_item_vec, n_item_vec))
    return out


def pointwise_step_of(user_vec, item_vec, item_bias, label, a, b):
    rows = (user_vec, item_vec, item_bias)
    if not all(isinstance(r, GatheredRows) for r in rows):
        return None
    U, V, bb = user_vec.factor, item_vec.factor, item_bias.factor
    if bb.dim != 1 or U.dim != V.dim or bb.num_instances != V.num_instances or not _same_ids(item_vec.ids, item_bias.ids):
        return None
    out = composed_model("wrmf", U, V, bb, a=float(a), b=float(b))(user_vec.ids, item_vec.ids, label)
    _tag(out[0]._step, (user_vec, item_vec))
    return out


class L2Sum:
    """`tf.nn.l2_loss(rows) + tf.nn.l2_loss(rows) + ...` of looked-up vectors: the second output of the fused step whose
    lookups they are (bpr.py:35, wrmf.py:32), a host sum otherwise"""

    def __init__(self, rows, extra=0.0):
        self.rows, self.extra = list(rows), extra
        self._step = None

    def __add__(self, other):
        if isinstance(other, L2Sum):
            return L2Sum(self.rows + other.rows, self.extra + other.extra)
        return L2Sum(self.rows, self.extra + float(other))

    __radd__ = __add__

    def resolve(self):
        """the recorded step whose l2 term this is: the lookups of ONE loss-module call, each once, nothing added"""
        if self._step is None and self.extra == 0.0:
            st = getattr(self.rows[0], "_l2_step", None)
            want = getattr(st, "l2_rows", ())
            if st is not None and len(want) == len(self.rows) and all(any(r is w for r in self.rows) for w in want):
                self._step = st
        return self._step

    def numpy(self):
        st = self.resolve()
        if st is not None:
            return np.float32(st.forward()[1])
        return np.float32(sum(0.5 * float((np.asarray(r, np.float64) ** 2).sum()) for r in self.rows) + self.extra)

    def __float__(self):
        return float(self.numpy())

    def __array__(self, dtype=None, copy=None):
        a = np.asarray(self.numpy())
        return a.astype(dtype) if dtype is not None else a

    def __format__(self, spec):
        return format(float(self), spec)


def l2_loss(x):
    """tf.nn.l2_loss: sum(x ** 2) / 2"""
    if isinstance(x, GatheredRows):
        return L2Sum([x])
    return np.float32(0.5 * float((np.asarray(x, np.float64) ** 2).sum()))
