"""A stand-in `tensorflow` module on torch CPU autograd -- TEST INFRASTRUCTURE, not product.

Purpose: TensorFlow 2.0.1 (docs_requirements.txt:2 of the reference) is not installable in the build container, so the
reference's own model code cannot run as it is.  This module provides exactly the TensorFlow symbols that
`openrec/tf2/{modules,recommenders,metrics}` touch, so that the reference's OWN FILES -- imported
from /root/reference, never copied -- execute on the CPU: the graph of every recommender (gathers, reductions, losses) is then
the reference's text, differentiated by torch autograd, and only what TensorFlow itself does is restated here:

  * op semantics (`tf.maximum` sends the gradient to its first argument on ties, `tf.nn.l2_loss = sum(x^2)/2`,
    `tf.unique` in first-occurrence order, `LinearOperatorLowerTriangular(..).to_dense()`, `band_part`, Keras losses);
  * embedding-lookup gradients as IndexedSlices (one value row per occurrence, the lookups of a variable concatenated);
  * the Keras OptimizerV2 sparse rules of TF 2.0.x: SGD scatter-adds every occurrence, Adagrad and Adam
    first sum duplicate indices (unique + segment sum), Adam's sparse apply decays m, v and moves var over the WHOLE table.

`install(dtype)` registers the module tree in sys.modules (tensorflow, tensorflow.keras, tensorflow.keras.layers, ...).
tests/golden/make_golden_tf.py uses it as its dry-run backend (`--backend stub`); with a real TensorFlow 2.0.1 the same
script runs on the real thing (`--backend tf`).
"""
from __future__ import annotations

import sys
import types

import numpy as np
import torch

_DT = torch.float64          # floating-point type of the run (install(dtype))
_TAPES = []                  # active GradientTapes (innermost last)


# ------------------------------------------------------------------ tensors and variables ---
def _t(x, dtype=None):
    """anything -> torch tensor (Variables: their live value, inside the autograd graph)"""
    if isinstance(x, Variable):
        return x.t
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(dtype)
    a = np.asarray(x)
    if dtype is None:
        dtype = _DT if a.dtype.kind == "f" else (torch.bool if a.dtype.kind == "b" else torch.int64)
    return torch.as_tensor(a).to(dtype)


def _np(x):
    return _t(x).detach().cpu().numpy()


class Variable:
    """tf.Variable: a leaf tensor; arithmetic on it goes through its live value"""

    def __init__(self, initial_value, dtype=None, trainable=True, name=None):
        self.t = _t(initial_value, dtype).detach().clone().requires_grad_(_t(initial_value, dtype).is_floating_point())
        self.name, self.trainable = name, trainable

    @property
    def shape(self):
        return tuple(self.t.shape)

    def numpy(self):
        return self.t.detach().cpu().numpy()

    def _set(self, new):
        with torch.no_grad():
            self.t.copy_(_t(new, self.t.dtype))
        return self

    assign = _set

    def assign_add(self, d):
        return self._set(self.t.detach() + _t(d, self.t.dtype))

    def scatter_nd_update(self, indices, updates):
        idx = _t(indices).reshape(-1).long()
        with torch.no_grad():
            self.t[idx] = _t(updates, self.t.dtype).detach()
        return self

    def scatter_add_rows(self, idx, rows):          # (optimizers)
        with torch.no_grad():
            self.t.index_add_(0, idx, rows.to(self.t.dtype))

    # arithmetic: the reference writes `x - variables[0]`, `matmul(x, variables[0])`, `reshape(variables[0], ..)`
    def __add__(self, o): return self.t + _t(o)
    def __radd__(self, o): return _t(o) + self.t
    def __sub__(self, o): return self.t - _t(o)
    def __rsub__(self, o): return _t(o) - self.t
    def __mul__(self, o): return self.t * _t(o)
    def __rmul__(self, o): return _t(o) * self.t
    def __truediv__(self, o): return self.t / _t(o)
    def __neg__(self): return -self.t


class IndexedSlices:
    def __init__(self, values, indices, dense_shape):
        self.values, self.indices, self.dense_shape = values, indices, dense_shape

    def to_dense(self):
        out = torch.zeros(self.dense_shape, dtype=self.values.dtype)
        out.index_add_(0, self.indices, self.values)
        return out


class GradientTape:
    def __enter__(self):
        self.lookups = []            # (variable, ids, leaf standing for the gathered rows)
        _TAPES.append(self)
        return self

    def __exit__(self, *exc):
        _TAPES.remove(self)
        return False

    def gradient(self, target, sources):
        # a nested target is differentiated as the sum of its elements (tf.GradientTape.gradient; tf2_examples/bpr_citeulike.py:35-37)
        flat = []
        def walk(x):
            if isinstance(x, (tuple, list)):
                for y in x:
                    walk(y)
            else:
                flat.append(_t(x).sum())
        walk(target)
        total = sum(flat)
        srcs = list(sources)
        leaves = [lk[2] for lk in self.lookups] + [v.t for v in srcs]
        g = torch.autograd.grad(total, leaves, allow_unused=True)
        nl = len(self.lookups)
        out = []
        for k, v in enumerate(srcs):
            vals = [(g[i], self.lookups[i][1]) for i in range(nl) if self.lookups[i][0] is v and g[i] is not None]
            dense = g[nl + k]
            if vals:
                assert dense is None, "variable used both through lookups and densely under one tape"
                out.append(IndexedSlices(torch.cat([a.reshape(-1, v.t.shape[-1]) for a, _ in vals]),
                                         torch.cat([i.reshape(-1) for _, i in vals]), tuple(v.t.shape)))
            else:
                out.append(dense)
        return out


def _lookup(var, ids):
    """embedding_lookup / gather on a variable's rows; under a tape the gathered rows are a leaf of their own, so that the
    gradient comes back per OCCURRENCE (IndexedSlices) as in TensorFlow"""
    ids = _t(ids).long()
    if _TAPES and isinstance(var, Variable):
        rows = var.t.detach()[ids].clone().requires_grad_(True)
        _TAPES[-1].lookups.append((var, ids, rows))
        return rows
    return _t(var)[ids]


# ------------------------------------------------------------------------------- ops ---
def _axis_kw(axis, keepdims):
    return {} if axis is None else dict(dim=axis, keepdim=bool(keepdims))


def reduce_sum(x, axis=None, keepdims=False, name=None):
    x = _t(x)
    return x.sum() if axis is None else x.sum(dim=axis, keepdim=bool(keepdims))


def reduce_mean(x, axis=None, keepdims=False, name=None):
    x = _t(x)
    return x.mean() if axis is None else x.mean(dim=axis, keepdim=bool(keepdims))


def maximum(x, y, name=None):
    # the gradient of Maximum goes to x where x >= y (math_grad.py: _MaximumMinimumGrad with greater_equal), never split
    x, y = _t(x), _t(y, None if not isinstance(y, (int, float)) else _t(x).dtype)
    x, y = torch.broadcast_tensors(x, y)
    return torch.where(x >= y, x, y)


def minimum(x, y, name=None):
    x, y = _t(x), _t(y, None if not isinstance(y, (int, float)) else _t(x).dtype)
    x, y = torch.broadcast_tensors(x, y)
    return torch.where(x <= y, x, y)


def clip_by_value(x, lo, hi, name=None):
    return minimum(maximum(x, lo), hi)          # clip_ops.clip_by_value: minimum(maximum(t, min), max)


def log_sigmoid(x, name=None):
    return -torch.nn.functional.softplus(-_t(x))


def l2_loss(x, name=None):
    x = _t(x)
    return (x * x).sum() / 2


def unique(x):
    a = _np(x).reshape(-1)
    _, first = np.unique(a, return_index=True)
    vals = a[np.sort(first)]                    # first-occurrence order (tf.unique)
    pos = {int(v): k for k, v in enumerate(vals)}
    return torch.as_tensor(vals), torch.as_tensor(np.array([pos[int(v)] for v in a], np.int64))


def gather(params, indices, axis=0, name=None):
    assert axis == 0
    return _lookup(params, indices)


def norm(x, axis=None, keepdims=False, ord="euclidean"):
    x = _t(x)
    return torch.sqrt((x * x).sum(**_axis_kw(axis, keepdims)))


def matmul(a, b, transpose_a=False, transpose_b=False, name=None):
    a, b = _t(a), _t(b)
    if transpose_a:
        a = a.transpose(-1, -2)
    if transpose_b:
        b = b.transpose(-1, -2)
    return a @ b


def band_part(x, num_lower, num_upper, name=None):
    x = _t(x)
    n, m = x.shape[-2], x.shape[-1]
    i, j = torch.arange(n).reshape(-1, 1), torch.arange(m).reshape(1, -1)
    keep = ((num_lower < 0) | (i - j <= num_lower)) & ((num_upper < 0) | (j - i <= num_upper))
    return x * keep.to(x.dtype)


class LinearOperatorLowerTriangular:
    def __init__(self, tril):
        self._m = _t(tril)

    def to_dense(self):
        return torch.tril(self._m)          # the operator ignores what stands above the diagonal


def boolean_mask(tensor, mask, name=None):
    m = _t(mask)
    return _t(tensor)[m if m.dtype == torch.bool else m != 0]


def reshape(x, shape, name=None):
    shape = [int(s) for s in (shape if isinstance(shape, (list, tuple)) else list(_np(shape)))]
    return _t(x).reshape(shape)


def shape(x):
    return tuple(_t(x).shape)


def size(x):
    return int(_t(x).numel())


def cast(x, dtype):
    return _t(x).to(dtype)


def constant(v, dtype=None, shape=None, name=None):
    return _t(v, dtype)


def zeros(shape, dtype=None):
    return torch.zeros(tuple(int(s) for s in shape) if not isinstance(shape, int) else (shape,), dtype=dtype or _DT)


def map_fn(fn, elems, dtype=None, parallel_iterations=None):
    cols = [_t(e) for e in elems] if isinstance(elems, (tuple, list)) else None
    n = cols[0].shape[0] if cols else _t(elems).shape[0]
    res = [fn(tuple(c[i] for c in cols)) if cols else fn(_t(elems)[i]) for i in range(n)]
    return torch.stack([_t(r) for r in res])


def count_nonzero(x, axis=None, keepdims=False, dtype=torch.int64):
    nz = _t(x) != 0
    r = nz.sum() if axis is None else nz.sum(dim=axis, keepdim=bool(keepdims))
    return r.to(dtype)


# -------------------------------------------------------------------------- keras layers ---
class Layer:
    def __init__(self, name=None, **kw):
        object.__setattr__(self, "_tracked", [])
        self.name = name

    def __setattr__(self, k, v):
        tracked = self.__dict__.get("_tracked")
        if tracked is None:                      # (a subclass that sets attributes before calling Layer.__init__)
            object.__setattr__(self, "_tracked", [])
            tracked = self._tracked
        if isinstance(v, Layer) or (isinstance(v, (list, tuple)) and v and all(isinstance(e, Layer) for e in v)):
            tracked.append(v)
        object.__setattr__(self, k, v)

    def __call__(self, *args, **kw):
        return self.call(*args, **kw)

    def _own_variables(self):
        return []

    @property
    def variables(self):
        out = list(self._own_variables())
        for t in self._tracked:
            for l in (t if isinstance(t, (list, tuple)) else [t]):
                out += l.variables
        return out

    @property
    def trainable_variables(self):
        return [v for v in self.variables if v.trainable]

    weights = variables

    def get_weights(self):
        return [v.numpy() for v in self.variables]

    def set_weights(self, ws):
        vs = self.variables
        assert len(vs) == len(ws), (len(vs), len(ws))
        for v, w in zip(vs, ws):
            assert tuple(v.shape) == tuple(np.shape(w)), (v.shape, np.shape(w))
            v.assign(w)


class Model(Layer):
    pass


def _init(kind, shape, rng):
    if kind == "zeros":
        return np.zeros(shape)
    if kind == "uniform":                        # keras 'uniform' = RandomUniform(-0.05, 0.05)
        return rng.uniform(-0.05, 0.05, shape)
    if kind == "glorot_uniform":
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, shape)
    raise ValueError(kind)


_RNG = np.random.default_rng(0)


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, embeddings_initializer="uniform", name=None, **kw):
        super().__init__(name=name)
        self.embeddings = Variable(_init(embeddings_initializer, (input_dim, output_dim), _RNG), _DT, name=name)

    def _own_variables(self):
        return [self.embeddings]

    def __call__(self, ids):
        return _lookup(self.embeddings, ids)


_ACT = {None: lambda x: x, "linear": lambda x: x, "relu": torch.relu, "sigmoid": torch.sigmoid}


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, name=None, **kw):
        super().__init__(name=name)
        self.units, self.activation, self.use_bias, self.kernel, self.bias = units, activation, use_bias, None, None

    def _own_variables(self):
        return [v for v in (self.kernel, self.bias) if v is not None]

    def __call__(self, x):
        x = _t(x)
        if self.kernel is None:                  # built on first call (glorot-uniform kernel, zero bias)
            self.kernel = Variable(_init("glorot_uniform", (x.shape[-1], self.units), _RNG), _DT)
            if self.use_bias:
                self.bias = Variable(np.zeros(self.units), _DT)
        y = x @ self.kernel.t
        if self.use_bias:
            y = y + self.bias.t
        return _ACT[self.activation](y)


class Sequential(Model):
    def __init__(self, layers=None, name=None):
        super().__init__(name=name)
        self.layers = []
        for l in layers or []:
            self.add(l)

    def add(self, layer):
        self.layers.append(layer)
        self._tracked.append(layer)

    def __call__(self, x):
        for l in self.layers:
            x = l(x)
        return x


class _Loss:
    def __call__(self, y_true, y_pred, sample_weight=None):
        return self.call(_t(y_true, _DT), _t(y_pred))


class MeanSquaredError(_Loss):
    def call(self, y, p):
        return ((p - y) ** 2).mean()             # mean over the last axis, then over the batch: the mean of a [B] vector


class BinaryCrossentropy(_Loss):
    def __init__(self, from_logits=False, **kw):
        self.from_logits = from_logits

    def call(self, y, p):
        if self.from_logits:                     # nn.sigmoid_cross_entropy_with_logits: max(z,0) - z y + log(1 + exp(-|z|))
            return (torch.clamp(p, min=0) - p * y + torch.log1p(torch.exp(-p.abs()))).mean()
        eps = 1e-7                               # backend.binary_crossentropy: clip to [eps, 1-eps], log(p + eps)
        pc = clip_by_value(p, eps, 1 - eps)
        return (-(y * torch.log(pc + eps) + (1 - y) * torch.log(1 - pc + eps))).mean()


class Mean:
    def __init__(self): self.reset_states()
    def reset_states(self): self.s, self.n = 0.0, 0
    def update_state(self, v):
        a = np.concatenate([np.asarray(_np(x), np.float64).reshape(-1) for x in (v if isinstance(v, (tuple, list)) else [v])])
        self.s += float(a.sum()); self.n += a.size
    def result(self): return torch.tensor(self.s / max(self.n, 1))


# ---------------------------------------------------------------------- keras optimizers ---
class _Optimizer:
    """OptimizerV2, sparse path.  Dense gradients take the dense rules of the same optimizers."""
    _slot_names = ()

    def __init__(self):
        self._slots, self.iterations = {}, 0

    def get_slot(self, var, name):
        return self._slots[(id(var), name)]

    def _slot(self, var, name, init=0.0):
        key = (id(var), name)
        if key not in self._slots:
            self._slots[key] = Variable(np.full(var.shape, init), var.t.dtype, trainable=False)
        return self._slots[key]

    def apply_gradients(self, grads_and_vars):
        self.iterations += 1
        for g, v in grads_and_vars:
            if g is None:
                continue
            if isinstance(g, IndexedSlices):
                if self._dedup:
                    uq, inv = torch.unique(g.indices, return_inverse=True)          # unique + unsorted_segment_sum
                    summed = torch.zeros((uq.numel(),) + tuple(g.values.shape[1:]), dtype=g.values.dtype)
                    summed.index_add_(0, inv, g.values)
                    self._sparse(v, summed, uq)
                else:
                    self._sparse(v, g.values, g.indices)
            else:
                self._dense(v, g)


class SGD(_Optimizer):
    _dedup = False                               # gradient_descent.py overrides _resource_apply_sparse_duplicate_indices

    def __init__(self, learning_rate=0.01, **kw):
        super().__init__(); self.lr = learning_rate

    def _sparse(self, v, vals, idx):
        v.scatter_add_rows(idx, -self.lr * vals)

    def _dense(self, v, g):
        v.assign(v.t.detach() - self.lr * g)


class Adagrad(_Optimizer):
    _dedup = True

    def __init__(self, learning_rate=0.001, initial_accumulator_value=0.1, epsilon=1e-7, **kw):
        super().__init__(); self.lr, self.init, self.eps = learning_rate, initial_accumulator_value, epsilon

    def _sparse(self, v, G, idx):
        acc = self._slot(v, "accumulator", self.init)
        with torch.no_grad():
            a = acc.t[idx] + G * G
            acc.t[idx] = a
            v.t[idx] = v.t[idx] - self.lr * G / (torch.sqrt(a) + self.eps)

    def _dense(self, v, g):
        acc = self._slot(v, "accumulator", self.init)
        acc.assign(acc.t.detach() + g * g)
        v.assign(v.t.detach() - self.lr * g / (torch.sqrt(acc.t.detach()) + self.eps))


class Adam(_Optimizer):
    _dedup = True

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **kw):
        super().__init__(); self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon

    def _lr_t(self):
        t = self.iterations
        return self.lr * np.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)

    def _sparse(self, v, G, idx):
        # adam.py (TF 2.0.x) _resource_apply_sparse: m, v decayed over the whole variable, the slices scattered in, var moved everywhere
        m, vv = self._slot(v, "m"), self._slot(v, "v")
        with torch.no_grad():
            m.t.mul_(self.b1); m.t.index_add_(0, idx, (1 - self.b1) * G)
            vv.t.mul_(self.b2); vv.t.index_add_(0, idx, (1 - self.b2) * G * G)
            v.t.sub_(self._lr_t() * m.t / (torch.sqrt(vv.t) + self.eps))

    def _dense(self, v, g):
        m, vv = self._slot(v, "m"), self._slot(v, "v")
        with torch.no_grad():
            m.t.mul_(self.b1).add_((1 - self.b1) * g)
            vv.t.mul_(self.b2).add_((1 - self.b2) * g * g)
            v.t.sub_(self._lr_t() * m.t / (torch.sqrt(vv.t) + self.eps))


# ------------------------------------------------------------------------------ install ---
def install(dtype="float64"):
    """Register the stand-in as `tensorflow` (and the submodules the reference imports from).  Returns the module."""
    global _DT
    _DT = {"float64": torch.float64, "float32": torch.float32}[dtype]
    tf = types.ModuleType("tensorflow")
    tf.__version__ = "2.0.1-stub(torch %s, %s)" % (torch.__version__, dtype)
    tf.int32, tf.int64, tf.float32, tf.float64, tf.bool = torch.int64, torch.int64, _DT, torch.float64, torch.bool
    tf.Variable, tf.GradientTape, tf.IndexedSlices = Variable, GradientTape, IndexedSlices
    tf.function = lambda f=None, **kw: (f if f is not None else (lambda g: g))
    for name, fn in dict(reduce_sum=reduce_sum, reduce_mean=reduce_mean, maximum=maximum, minimum=minimum, clip_by_value=clip_by_value,
                         unique=unique, gather=gather, norm=norm, matmul=matmul, boolean_mask=boolean_mask, reshape=reshape, shape=shape,
                         size=size, cast=cast, constant=constant, zeros=zeros, map_fn=map_fn,
                         expand_dims=lambda x, axis, name=None: _t(x).unsqueeze(axis),
                         squeeze=lambda x, axis=None, name=None: _t(x).squeeze() if axis is None else _t(x).squeeze(axis),
                         stack=lambda xs, axis=0, name=None: torch.stack([_t(x) for x in xs], dim=axis),
                         unstack=lambda x, axis=0, name=None: list(torch.unbind(_t(x), dim=axis)),
                         concat=lambda xs, axis, name=None: torch.cat([_t(x) for x in xs], dim=axis),
                         tile=lambda x, m, name=None: _t(x).repeat(*[int(k) for k in m]),
                         ones_like=lambda x, dtype=None, name=None: torch.ones_like(_t(x)),
                         square=lambda x, name=None: _t(x) ** 2).items():
        setattr(tf, name, fn)
    m = types.ModuleType("tensorflow.math")
    for name, fn in dict(reduce_sum=reduce_sum, reduce_mean=reduce_mean, log_sigmoid=log_sigmoid, maximum=maximum, minimum=minimum,
                         count_nonzero=count_nonzero,
                         multiply=lambda a, b, name=None: _t(a) * _t(b), sigmoid=lambda x, name=None: torch.sigmoid(_t(x)),
                         square=lambda x, name=None: _t(x) ** 2, exp=lambda x, name=None: torch.exp(_t(x)),
                         log=lambda x, name=None: torch.log(_t(x, _DT)), reciprocal=lambda x, name=None: 1.0 / _t(x),
                         logical_not=lambda x, name=None: ~_t(x).bool(), logical_or=lambda a, b, name=None: _t(a).bool() | _t(b).bool()).items():
        setattr(m, name, fn)
    nn = types.ModuleType("tensorflow.nn"); nn.l2_loss = l2_loss
    nn.embedding_lookup = lambda params, ids, name=None: _lookup(params, ids)
    la = types.ModuleType("tensorflow.linalg")
    la.matmul, la.band_part, la.LinearOperatorLowerTriangular = matmul, band_part, LinearOperatorLowerTriangular
    keras = types.ModuleType("tensorflow.keras")
    layers = types.ModuleType("tensorflow.keras.layers"); layers.Layer, layers.Embedding, layers.Dense = Layer, Embedding, Dense
    losses = types.ModuleType("tensorflow.keras.losses"); losses.MeanSquaredError, losses.BinaryCrossentropy = MeanSquaredError, BinaryCrossentropy
    opts = types.ModuleType("tensorflow.keras.optimizers"); opts.SGD, opts.Adagrad, opts.Adam = SGD, Adagrad, Adam
    metrics = types.ModuleType("tensorflow.keras.metrics"); metrics.Mean = Mean
    keras.Model, keras.Sequential = Model, Sequential
    keras.layers, keras.losses, keras.optimizers, keras.metrics = layers, losses, opts, metrics
    tf.math, tf.nn, tf.linalg, tf.keras = m, nn, la, keras
    for mod in (tf, m, nn, la, keras, layers, losses, opts, metrics):
        sys.modules[mod.__name__] = mod
    return tf


def uninstall():
    for k in [k for k in sys.modules if k == "tensorflow" or k.startswith("tensorflow.")]:
        del sys.modules[k]
    for k in [k for k in sys.modules if k == "openrec" or k.startswith("openrec.")]:
        del sys.modules[k]
