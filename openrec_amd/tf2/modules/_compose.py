"""Module-level compositions that run as ONE fused device step.

The reference builds its recommenders out of modules (recommenders/bpr.py:21-37, wrmf.py:21-34): `LatentFactor` lookups,
a loss module on the looked-up vectors, `tf.nn.l2_loss` of the same vectors, then `tape.gradient((loss, l2_loss), vars)` and
`optimizer.apply_gradients` (tf2_examples/bpr_citeulike.py:33-39).  A user who writes such a model by hand gets the same
fused kernel as `openrec_amd.tf2.recommenders.BPR / WRMF`: the loss modules recognise lookups of three (two) tables and
record a pending step on a model object made of those very `LatentFactor`s; the l2 terms resolve to the step's second
output.  What is NOT a recognised composition has no gradients (it raises under a tape); looked at outside a tape, its MLP and
interaction nodes run on the device (orx_mlp_forward / orx_interact_forward) and only element-wise glue on fetched values is NumPy."""
from __future__ import annotations

import warnings
import weakref

import numpy as np

from .latent_factor import GatheredRows

_models = weakref.WeakValueDictionary()     # introspection only: the strong reference lives on the user LatentFactor
_warned = set()


def host_fallback(what):
    from .._lazy import active_tape
    if active_tape() is not None:
        # under a tape the caller expects to TRAIN through this value: a host forward has no gradients
        raise NotImplementedError(f"{what}: this call is not one of the compositions this package runs as a fused device step (bpr.py / ucml.py "
                                  "/ gmf.py / wrmf.py / dlrm.py of the reference) -- it would be computed on the host WITHOUT gradients; evaluate "
                                  "it outside the GradientTape if only its value is wanted")
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
    mlp = kw.pop("mlp", None)
    key = (kind, id(user_lf), id(item_lf), id(bias_lf), id(mlp), tuple(sorted(kw.items())))
    owned = user_lf.__dict__.setdefault("_composed", {})
    m = owned.get(key)
    if m is None:
        if kind == "bpr":
            class _Composed(PairwiseRecommender):
                _model = "bpr"
        elif kind == "ucml":
            class _Composed(PairwiseRecommender):
                _model = "ucml"
                _score_kind = "l2"
                margin = float(kw["margin"])
        elif kind == "gmf":
            class _Composed(PointwiseRecommender):
                _score_kind = "gmf"

                def _point_args(self):
                    return "gmf", self.mlp.layers[0].kernel, {}
        else:
            class _Composed(PointwiseRecommender):
                def _point_args(self):
                    return "wrmf", None, dict(a=kw["a"], b_w=kw["b"], sigmoid=kw.get("sigmoid", False))
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
        if mlp is not None:
            m.mlp = mlp
            _chain(mlp.layers[0].kernel, flush)
        owned[key] = m
        _models[key] = m
    return m


def _tag(step, rows, params=()):
    step.l2_rows = rows
    step.l2_params = tuple(params)
    for r in rows:
        r._l2_step = step


def _consume(*rows):
    for r in rows:
        r.consumed()


# ---- compositions spelled out in raw ops (modules/_expr.py) ------------------------------------------------------------
def _node(e, op, n=None):
    from ._expr import Expr
    return isinstance(e, Expr) and e.op == op and (n is None or len(e.args) == n)


def _sq_dist(e, keepdims):
    """reduce_sum(square(a - b), axis=-1, keepdims) -> (a, b)"""
    if not (_node(e, "reduce_sum", 1) and e.kw.get("axis") in (-1,) and bool(e.kw.get("keepdims")) == keepdims):
        return None
    sq = e.args[0]
    if not (_node(sq, "square", 1) and _node(sq.args[0], "sub", 2)):
        return None
    return sq.args[0].args


def _ucml_score(e):
    """(-|u - v|^2) + bias  (ucml.py:35-36)  -> (u rows, item rows, bias rows)"""
    if not _node(e, "add", 2):
        return None
    a, b = e.args
    if isinstance(a, GatheredRows):
        a, b = b, a
    if not (_node(a, "neg", 1) and isinstance(b, GatheredRows)):
        return None
    d = _sq_dist(a.args[0], True)
    if d is None or not all(isinstance(x, GatheredRows) for x in d):
        return None
    return d[0], d[1], b


def try_ucml_loss(e):
    """tf.reduce_sum(tf.maximum(margin - (pos_score - neg_score), 0))  (ucml.py:37-39) -> the fused UCML step's loss, or None"""
    mx = e.args[0]
    if not _node(mx, "maximum", 2):
        return None
    h, zero = mx.args
    if not (np.isscalar(zero) and float(zero) == 0.0 and _node(h, "sub", 2) and np.isscalar(h.args[0]) and _node(h.args[1], "sub", 2)):
        return None
    pos, neg = _ucml_score(h.args[1].args[0]), _ucml_score(h.args[1].args[1])
    if pos is None or neg is None:
        return None
    (u1, p, bp), (u2, n, bn) = pos, neg
    U, V, b = u1.factor, p.factor, bp.factor
    if u2 is not u1 and not (u2.factor is U and _same_ids(u1.ids, u2.ids)):
        return None
    if n.factor is not V or bn.factor is not b or b.dim != 1 or U.dim != V.dim or b.num_instances != V.num_instances:
        return None
    if not (_same_ids(p.ids, bp.ids) and _same_ids(n.ids, bn.ids)):
        return None
    _consume(u1, u2, p, n, bp, bn)
    out = composed_model("ucml", U, V, b, margin=float(h.args[0]))(u1.ids, p.ids, n.ids)
    _tag(out[0]._step, (u1, p, n))
    return out[0]


def _gmf_logit(e):
    """reshape(mlp(u * i) + b, [-1])  (gmf.py:28) -> (mlp, u rows, item rows, bias rows)"""
    if _node(e, "reshape", 1):
        e = e.args[0]
    if not _node(e, "add", 2):
        return None
    a, b = e.args
    if isinstance(a, GatheredRows):
        a, b = b, a
    if not (_node(a, "dense1", 2) and isinstance(b, GatheredRows) and _node(a.args[1], "mul", 2)):
        return None
    u, i = a.args[1].args
    if not (isinstance(u, GatheredRows) and isinstance(i, GatheredRows)):
        return None
    return a.args[0], u, i, b


def try_gmf_loss(label, logit):
    """BinaryCrossentropy(from_logits=True)(label, reshape(mlp(u * i) + b_i))  (gmf.py:28-29) -> the fused GMF step's loss, or None"""
    g = _gmf_logit(logit)
    if g is None:
        return None
    mlp, u, i, b = g
    U, V, bb = u.factor, i.factor, b.factor
    if bb.dim != 1 or U.dim != V.dim or bb.num_instances != V.num_instances or not _same_ids(i.ids, b.ids):
        return None
    _consume(u, i, b)
    out = composed_model("gmf", U, V, bb, mlp=mlp)(u.ids, i.ids, label)
    _tag(out[0]._step, (u, i), params=(mlp.layers[0].kernel,))
    return out[0]


def try_all_item_scores(e):
    """the two inference compositions that end in `+ tf.reshape(item_bias.variables[0], [-1])`:
    ucml.py:50-53  -reduce_sum(square(expand_dims(user_vec, 1) - V), -1)   and   gmf.py:36-41  squeeze(mlp(expand_dims(user_vec, 1) * V), -1)"""
    from ... import runtime as rt
    from .latent_factor import Variable
    from ._expr import Expr

    def flat_var(x):            # tf.reshape(variable, [-1]) (compat._Flat); never touches a lookup (that would gather it)
        return None if isinstance(x, (GatheredRows, Expr, Variable)) or np.isscalar(x) or isinstance(x, np.ndarray) else getattr(x, "flat_of", None)
    a, b = e.args
    bias = flat_var(b) or flat_var(a)
    body = a if flat_var(b) is not None else b
    if not isinstance(bias, Variable) or bias.table.dim != 1:
        return None

    def user_and_items(x, y):
        if _node(x, "expand_dims", 1) and x.kw.get("axis") == 1 and isinstance(x.args[0], GatheredRows) and isinstance(y, Variable):
            return x.args[0], y
        return None
    kind = w = ui = None
    if _node(body, "neg", 1):
        d = _sq_dist(body.args[0], False)
        ui = user_and_items(*d) if d is not None else None
        kind = "l2"
    elif _node(body, "squeeze", 1) and _node(body.args[0], "dense1", 2) and _node(body.args[0].args[1], "mul", 2):
        ui = user_and_items(*body.args[0].args[1].args)
        kind, w = "gmf", body.args[0].args[0].layers[0].kernel
    if ui is None:
        return None
    rows, item_var = ui
    if item_var.table.rows != bias.table.rows or rows.factor.dim != item_var.table.dim:
        return None
    rows.consumed()          # (the scorer reads the user rows in HBM: nobody will look at this lookup on the host)
    return rt.score_all_items(kind, rows.factor.table, item_var.table, bias.table, rows.flat_ids(), w=w, device=True)


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
    _consume(*rows)
    out = composed_model("bpr", U, V, b)(user_vec.ids, p_item_vec.ids, n_item_vec.ids)
    _tag(out[0]._step, (user_vec, p_item_vec, n_item_vec))
    return out


def pointwise_step_of(user_vec, item_vec, item_bias, label, a, b, sigmoid=False):
    rows = (user_vec, item_vec, item_bias)
    if not all(isinstance(r, GatheredRows) for r in rows):
        return None
    U, V, bb = user_vec.factor, item_vec.factor, item_bias.factor
    if bb.dim != 1 or U.dim != V.dim or bb.num_instances != V.num_instances or not _same_ids(item_vec.ids, item_bias.ids):
        return None
    _consume(*rows)
    out = composed_model("wrmf", U, V, bb, a=float(a), b=float(b), sigmoid=bool(sigmoid))(user_vec.ids, item_vec.ids, label)
    _tag(out[0]._step, (user_vec, item_vec))
    return out


class L2Sum:
    """`tf.nn.l2_loss(rows) + tf.nn.l2_loss(rows) + ...` of looked-up vectors: the second output of the fused step whose
    lookups they are (bpr.py:35, wrmf.py:32), a host sum otherwise"""

    def __init__(self, rows, extra=0.0, params=()):
        self.rows, self.extra, self.params = list(rows), extra, list(params)      # params: dense kernels (gmf.py:32: the mlp's variables)
        self._step = None

    def __add__(self, other):
        mine = getattr(self, "scale", 1.0)
        if isinstance(other, L2Sum):
            theirs = getattr(other, "scale", 1.0)
            if mine == theirs:
                # `k * l2(a) + k * l2(b)`: the rows merge and the common weight travels with them
                out = L2Sum(self.rows + other.rows, self.extra + other.extra, self.params + other.params)
            else:
                # different weights: no fused step takes such a term -- the scaled VALUE goes into `extra` (a host number), and
                # resolve() then never matches a recorded step (tape.gradient raises instead of training with another weight)
                out = L2Sum(self.rows, self.extra + float(other.numpy()), self.params)
            if mine != 1.0:
                out.scale = mine
            return out
        out = L2Sum(self.rows, self.extra + float(other), self.params)
        if mine != 1.0:
            out.scale = mine
        return out

    __radd__ = __add__

    def __mul__(self, k):
        """`l2_reg * tf.nn.l2_loss(vec)`: the value is right, but the fused step takes the l2 term with weight 1 (bpr.py:35-37 sums it
        unscaled into the objective) or not at all -- tape.gradient says so instead of training with another weight"""
        out = L2Sum(self.rows, self.extra * float(k), self.params)
        out.scale = getattr(self, "scale", 1.0) * float(k)
        return out

    __rmul__ = __mul__

    def resolve(self):
        """the recorded step whose l2 term this is: the lookups of ONE loss-module call, each once, nothing added"""
        if self._step is None and self.extra == 0.0 and self.rows and getattr(self, "scale", 1.0) == 1.0:
            st = getattr(self.rows[0], "_l2_step", None)
            want, wantp = getattr(st, "l2_rows", ()), getattr(st, "l2_params", ())
            if st is not None and len(want) == len(self.rows) and all(any(r is w for r in self.rows) for w in want) \
                    and len(wantp) == len(self.params) and all(any(p is w for p in self.params) for w in wantp):
                self._step = st
        return self._step

    def numpy(self):
        st = self.resolve()
        if st is not None:
            return np.float32(st.forward()[1])
        return np.float32(getattr(self, "scale", 1.0) * (sum(0.5 * float((np.asarray(r, np.float64) ** 2).sum()) for r in self.rows)
                                                         + sum(0.5 * float((p.read().astype(np.float64) ** 2).sum()) for p in self.params)) + self.extra)

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
    from .latent_factor import Variable
    if isinstance(x, Variable):
        return L2Sum([], params=[x.table])
    return np.float32(0.5 * float((np.asarray(x, np.float64) ** 2).sum()))


# ---- DLRM: recommenders/dlrm.py:76-100 written against the modules --------------------------------------------------------
class _RowRangeTable:
    """Rows [row0, row0 + rows) of another table, as the `table` of a LatentFactor / Variable: the 26 embedding tables of the
    composition dlrm.py:30-31 are slices of the ONE combined table `orx_dlrm_step` trains (every access goes through that table:
    queued steps run first, a lazily-applied Adam is brought up to date)."""

    def __init__(self, base, row0, rows):
        self.base, self.row0, self.rows, self.dim, self.ctx = base, int(row0), int(rows), base.dim, base.ctx

    @property
    def shape(self):
        return (self.rows, self.dim)

    def read(self, row0=0, nrows=None):
        nrows = self.rows - row0 if nrows is None else nrows
        if row0 < 0 or row0 + nrows > self.rows:
            raise IndexError("row range outside the table")
        return self.base.read(self.row0 + row0, nrows)

    numpy = read

    def write(self, values, row0=0):
        a = np.ascontiguousarray(values, np.float32).reshape(-1, self.dim)
        if row0 < 0 or row0 + a.shape[0] > self.rows:
            raise IndexError("row range outside the table")
        self.base.write(a, self.row0 + row0)
        return self

    def gather(self, ids):
        i = np.asarray(ids).reshape(-1)
        if i.size and (i.min() < 0 or i.max() >= self.rows):
            raise IndexError("embedding id out of range")
        return self.base.gather((i.astype(np.int64) + self.row0).astype(np.int32))

    def censor(self, ids, min_norm=0.1):
        i = np.asarray(ids).reshape(-1)
        self.base.censor((i.astype(np.int64) + self.row0).astype(np.int32), min_norm)

    def fill(self, v):
        self.base.write(np.full((self.rows, self.dim), v, np.float32), self.row0)
        return self


def _match_dlrm(pred):
    """reshape([clip_by_value](mlp_top(concat([mlp_bot(dense), interaction(embeddings + [mlp_bot(dense)])], axis=1))), [-1])
    (dlrm.py:84-100) -> dict of its parts, or None"""
    from ._expr import Expr
    e = pred
    if not (_node(e, "reshape", 1) and list(np.atleast_1d(e.kw.get("shape"))) == [-1]):
        return None
    e = e.args[0]
    thr = 0.0
    if _node(e, "clip", 1):
        lo, hi = e.kw["lo"], e.kw["hi"]
        if not (0.0 < lo < 1.0 and abs((1.0 - lo) - hi) < 1e-6):
            return None
        thr, e = lo, e.args[0]
    if not _node(e, "mlp", 2):
        return None
    top, cat = e.args
    if not (_node(cat, "concat") and cat.kw.get("axis") == 1 and len(cat.args) == 2):
        return None
    bot_e, inter = cat.args
    if not (_node(bot_e, "mlp", 2) and _node(inter, "interact") and len(inter.args) >= 3):
        return None
    bot, dense = bot_e.args
    if isinstance(dense, (Expr, GatheredRows)) or inter.args[-1] is not bot_e:
        return None
    rows = list(inter.args[1:-1])
    if not all(isinstance(r, GatheredRows) and len(r.shape) == 2 for r in rows):
        return None
    lfs = [r.factor for r in rows]
    if len({id(f) for f in lfs}) != len(lfs) or len({f.dim for f in lfs}) != 1:
        return None
    if bot.layers[-1].units != lfs[0].dim or top.layers[-1].units != 1 or top is bot:
        return None
    acts = lambda m: ({l.activation for l in m.layers[:-1]} <= {"relu"}) and all(l.use_bias for l in m.layers) and m.layers[-1].activation in ("relu", "sigmoid")
    if not (acts(bot) and acts(top)):
        return None
    return dict(top=top, bot=bot, interaction=inter.args[0], dense=dense, rows=rows, lfs=lfs, threshold=thr)


def _dlrm_for(parts, loss_func):
    """the packaged DLRM recommender (recommenders/dlrm.py of this package: step queue, `orx_dlrm_step`) over THESE modules: made once
    per (latent factors, MLPs, interaction, loss), it takes the current values of the latent factors into its combined embedding
    table and from then on its parameters ARE the modules' (LatentFactor.table -> a row range of the combined table, Dense.kernel /
    .bias -> the model's parameter tables)"""
    from ..recommenders.dlrm import DLRM
    from .latent_factor import Variable
    top, bot, inter, lfs = parts["top"], parts["bot"], parts["interaction"], parts["lfs"]
    dense = np.asarray(parts["dense"])
    key = ("dlrm", tuple(id(f) for f in lfs), id(top), id(bot), id(inter), loss_func, float(parts["threshold"]))
    owned = lfs[0].__dict__.setdefault("_composed", {})
    m = owned.get(key)
    if m is not None:
        return m
    if any(isinstance(f.table, _RowRangeTable) for f in lfs) or any(getattr(l, "_adopted", False) for l in bot.layers + top.layers):
        # these modules already belong to another composition (another loss, another threshold ...): their parameters cannot be two models'
        raise NotImplementedError("DLRM composition: these modules are already the parameters of another fused DLRM step")
    ctx = lfs[0].table.ctx
    m = DLRM(lfs[0].dim, [f.num_instances for f in lfs], [l.units for l in bot.layers], [l.units for l in top.layers],
             arch_interaction_itself=inter._self_interaction, sigmoid_bot=bot.layers[-1].activation == "sigmoid",
             sigmoid_top=top.layers[-1].activation == "sigmoid", loss_func=loss_func, loss_threshold=parts["threshold"],
             reference_compat=inter._reference_compat, dense_dim=dense.shape[-1], ctx=ctx)
    emb = m._param("emb")
    row0 = 0
    for f in lfs:
        emb.write(f.table.read(), row0)                     # (the composition's embeddings keep their initial values)
        f.table = _RowRangeTable(emb, row0, f.num_instances)
        f._var = Variable(f.table, f._var.name)
        row0 += f.num_instances
    for name, mlp in (("bot", bot), ("top", top)):
        for l, layer in enumerate(mlp.layers):
            w, b = m._param(name + "_w", l), m._param(name + "_b", l)
            if layer.kernel is not None:                    # (built by an earlier host evaluation, e.g. inference before any loss call: keep those values)
                w.write(layer.kernel.read()); b.write(layer.bias.read())
            layer.kernel, layer.bias, layer._adopted = w, b, True
    owned[key] = m
    _models[key] = m
    return m


def _dlrm_sparse(parts):
    cols = []
    for r in parts["rows"]:
        i = r.flat_ids()
        if getattr(i, "is_cuda", False):
            i = i.cpu().numpy()
        cols.append(np.asarray(i).reshape(-1).astype(np.int32))
    return np.stack(cols, axis=1)


def try_dlrm_loss(loss_func, label, pred):
    """MeanSquaredError / BinaryCrossentropy()(y_true=label, y_pred=<the tree of dlrm.py:84-100>) (dlrm.py:72-73) -> the loss of the
    fused DLRM step, or None"""
    parts = _match_dlrm(pred)
    if parts is None:
        return None
    m = _dlrm_for(parts, loss_func)
    _consume(*parts["rows"])
    return m(np.asarray(parts["dense"], np.float32), _dlrm_sparse(parts), np.asarray(label, np.float32).reshape(-1))


def try_dlrm_inference(pred):
    """np.asarray(<the tree of dlrm.py:84-100>) (DLRM.inference, dlrm_criteo.py:52) -> predictions from `orx_dlrm_inference`, or None.
    Only for modules that already are a fused model's parameters (a loss call made them so)."""
    parts = _match_dlrm(pred)
    if parts is None:
        return None
    owned = parts["lfs"][0].__dict__.get("_composed", {})
    for key, m in owned.items():
        if key[0] == "dlrm" and key[1:5] == (tuple(id(f) for f in parts["lfs"]), id(parts["top"]), id(parts["bot"]), id(parts["interaction"])) \
                and key[6] == float(parts["threshold"]):
            _consume(*parts["rows"])
            return m.inference(np.asarray(parts["dense"], np.float32), _dlrm_sparse(parts))
    return None
