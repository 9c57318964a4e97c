"""LatentFactor: the embedding table of the reference
(openrec/tf2/modules/latent_factor.py:4-23: a Keras `Embedding` with a
`censor` method), backed by an HBM-resident `runtime.Table`."""
from __future__ import annotations

import numpy as np

from ... import runtime as rt

_seed_counter = [0]


class Variable:
    """What `layer.variables[0]` / `model.trainable_variables` hand out."""

    def __init__(self, table, name):
        self.table, self.name = table, name

    @property
    def shape(self):
        return self.table.shape

    def numpy(self):
        return self.table.read()

    def assign(self, value):
        self.table.write(np.asarray(value, np.float32))
        return self

    def __repr__(self):
        return f"<Variable {self.name} shape={self.table.shape} (HBM)>"


class GatheredRows:
    """What `LatentFactor.__call__` returns: the rows of `ids`, gathered from HBM when (and only when) somebody looks at them.

    The reference's recommenders are COMPOSITIONS of modules (bpr.py:23-33: five lookups, then `PairwiseLogLoss`, then
    `tf.nn.l2_loss` of the looked-up vectors).  Materialising the lookups on the host would take the composition off the
    device path -- 16 MB per gather at B = 65536 -- so a lookup stays a (table, ids) pair: `PairwiseLogLoss` /
    `PointwiseMSELoss` recognise their arguments and record ONE fused step (modules/_compose.py), exactly what
    `BPR.__call__` / `WRMF.__call__` of this package record.  Anything else that touches the object (np.asarray,
    arithmetic, indexing, `.numpy()`) gathers the rows to the host and continues there, as a plain array."""

    __array_priority__ = 100.0

    def __init__(self, factor, ids):
        self.factor = factor
        self.ids = ids
        self._host = None
        self._l2_step = None
        factor._pending.add(self)          # TF gathers at call time: see LatentFactor.snapshot_pending

    def flat_ids(self):
        i = self.ids
        if hasattr(i, "numpy") and not isinstance(i, np.ndarray) and not getattr(i, "is_cuda", False):
            i = i.numpy()
        return i if getattr(i, "is_cuda", False) else np.asarray(i)

    @property
    def shape(self):
        return tuple(np.shape(self.ids) if not hasattr(self.ids, "shape") else tuple(self.ids.shape)) + (self.factor.dim,)

    ndim = property(lambda self: len(self.shape))
    dtype = np.dtype(np.float32)

    def numpy(self):
        if self._host is None:
            i = self.flat_ids()
            if getattr(i, "is_cuda", False):
                i = i.cpu().numpy()
            self._host = self.factor.table.gather(np.asarray(i).reshape(-1)).reshape(self.shape)
            self.factor._pending.discard(self)
        return self._host

    def consumed(self):
        """a loss module turned this lookup into (part of) a fused step: nothing will ever look at its rows on the host"""
        self.factor._pending.discard(self)

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, k):
        return self.numpy()[k]

    def __getattr__(self, name):               # .sum(), .mean(), .reshape(), .T ... : the host array's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    def _bin(op, name=None):
        # with another lookup / variable / lazy expression the operation stays symbolic (modules/_expr.py: ucml.py:29-32 and
        # gmf.py:28 spell their scores out in raw ops); with a number or an array it is the host array's
        def f(self, other):
            if name is not None:
                from ._expr import Expr, is_lazy
                if is_lazy(other):
                    return Expr(name, self, other)
            return op(self.numpy(), np.asarray(other))
        def r(self, other):
            if name is not None:
                from ._expr import Expr, is_lazy
                if is_lazy(other):
                    return Expr(name, other, self)
            return op(np.asarray(other), self.numpy())
        return f, r

    __add__, __radd__ = _bin(np.add, "add")
    __sub__, __rsub__ = _bin(np.subtract, "sub")
    __mul__, __rmul__ = _bin(np.multiply, "mul")
    __truediv__, __rtruediv__ = _bin(np.true_divide)
    __matmul__, __rmatmul__ = _bin(np.matmul)
    del _bin

    def __neg__(self):
        return -self.numpy()

    def __repr__(self):
        return f"<GatheredRows {self.factor.name or 'latent_factor'}[{self.shape[0] if self.shape else ''}...] shape={self.shape} (lazy)>"


class LatentFactor:

    def __init__(self, num_instances, dim, zero_init=False, name=None, ctx=None, seed=None):
        self.num_instances, self.dim, self.name = int(num_instances), int(dim), name
        self.table = rt.Table(self.num_instances, self.dim, ctx)
        if zero_init:
            self.table.fill(0.0)                       # initializer 'zeros'  (latent_factor.py:8-9)
        else:
            if seed is None:
                _seed_counter[0] += 1
                seed = _seed_counter[0]
            self.table.init_uniform(-0.05, 0.05, seed)  # Keras 'uniform'      (latent_factor.py:10-11)
        self._var = Variable(self.table, (name or "latent_factor") + "/embeddings")
        import weakref
        self._pending = weakref.WeakSet()              # lookups nobody has looked at yet

    @property
    def variables(self):
        return [self._var]

    trainable_variables = variables

    def __call__(self, ids):
        """Embedding gather: [*] int ids -> [*, dim] fp32 -- lazily (see GatheredRows): `np.asarray(lf(ids))` is the host
        array, bit-exact rows of the table; an out-of-range id raises IndexError when the rows are gathered."""
        return GatheredRows(self, ids)

    call = __call__

    def snapshot_pending(self):
        """Gather every lookup that is still lazy NOW: called when a train step on this table is recorded.  TensorFlow gathers at
        call time, so `v = lf(ids); train_step(...); np.asarray(v)` must give the PRE-step rows; lookups that a loss module
        consumed (the five of bpr.py:23-27 ...) are not among them, so the fused path gathers nothing."""
        for r in list(self._pending):
            try:
                r.numpy()
            except IndexError:
                self._pending.discard(r)                 # (raised again when somebody looks at it)

    def censor(self, censor_id):
        """latent_factor.py:17-23: rows of the DISTINCT ids are divided by max(norm, 0.1)."""
        self.table.censor(censor_id, 0.1)
        return self._var
