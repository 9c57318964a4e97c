"""CPU oracle for the OpenRec tf2 embedding-training hot path (NumPy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (``openrec_amd/``) may
import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and there only as the checker.

PARITY UNPINNED: the reference (ylongqi/openrec) ships no tests, golden vectors
or stored outputs for this path, and its arithmetic lives in TensorFlow 2.0.1
(``docs_requirements.txt:2``), which is neither vendored under /root/reference
nor installable here.  This file therefore restates (a) the graph definitions
of the reference (cited file:line below, relative to /root/reference) and
(b) the published TF-2.0 semantics of the ops they call (gather, reduce,
log_sigmoid, l2_loss, GradientTape over a tuple target, IndexedSlices, Keras
OptimizerV2 sparse apply).  It is pinned three ways by ``tests/``:
analytic known-answer tests, an independent torch-CPU-autograd implementation
(``tests/golden/make_golden.py`` -> committed fixtures), and an fp64 "truth"
run of the same code.

Every function takes ``dtype`` (np.float32 = "as TF computes", np.float64 =
truth).  Tables are row-major ``[N, dim]`` arrays that are updated IN PLACE by
the ``*_step`` functions, exactly like Keras variables.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "log_sigmoid", "bpr_forward", "bpr_grads", "ucml_forward", "ucml_grads",
    "gmf_forward", "gmf_grads", "wrmf_forward", "wrmf_grads",
    "SGD", "Adagrad", "AdamTFSparse", "bpr_step", "ucml_step", "gmf_step",
    "wrmf_step", "censor", "tf_unique", "init_uniform",
    "bpr_inference", "ucml_inference",
]


# --------------------------------------------------------------------------
# parameters
# --------------------------------------------------------------------------
def init_uniform(rows, dim, seed, dtype=np.float32, lo=-0.05, hi=0.05):
    """Keras ``'uniform'`` initializer == RandomUniform(-0.05, 0.05)
    (openrec/tf2/modules/latent_factor.py:8-15).  The oracle uses NumPy's
    PCG64; the device RNG differs, so parity tests always *write* the oracle's
    tables into the device tables instead of relying on equal seeds."""
    rng = np.random.default_rng(seed)
    return rng.uniform(lo, hi, size=(rows, dim)).astype(dtype)


def tf_unique(ids):
    """``tf.unique`` returns values in first-occurrence order
    (latent_factor.py:19)."""
    ids = np.asarray(ids)
    _, first = np.unique(ids, return_index=True)
    return ids[np.sort(first)]


# --------------------------------------------------------------------------
# element-wise pieces
# --------------------------------------------------------------------------
def log_sigmoid(x):
    """``tf.math.log_sigmoid(x) = -softplus(-x)`` evaluated in a numerically
    stable form.  (TF's softplus switches between ``z``, ``exp(z)`` and
    ``log1p(exp(z))`` at +-(log(eps)+2); all three branches agree with this
    closed form to < 1 ulp of the result at fp32.)"""
    x = np.asarray(x)
    return -(np.maximum(-x, 0) + np.log1p(np.exp(-np.abs(x))))


def _sigmoid(x):
    x = np.asarray(x)
    e = np.exp(-np.abs(x))
    return np.where(x >= 0, 1 / (1 + e), e / (1 + e)).astype(x.dtype)


def _l2(*vs):
    """``tf.nn.l2_loss(t) = sum(t**2)/2`` (bpr.py:35)."""
    s = 0
    for v in vs:
        s = s + (v * v).sum(dtype=v.dtype) / 2
    return s


# --------------------------------------------------------------------------
# BPR  (openrec/tf2/recommenders/bpr.py:21-37, modules/pairwise_log_loss.py:15-34)
# --------------------------------------------------------------------------
def bpr_forward(U, V, b, uid, pid, nid):
    """Returns (loss, l2_loss, x) with x = pos_score - neg_score  [B]."""
    u, p, n = U[uid], V[pid], V[nid]                      # bpr.py:23-27 (ResourceGather)
    x = (u * p).sum(1) + b[pid, 0] - ((u * n).sum(1) + b[nid, 0])   # pairwise_log_loss.py:19-30
    m = np.maximum(x, np.asarray(-30.0, x.dtype))          # :32  tf.math.maximum(.., -30.0)
    loss = -log_sigmoid(m).mean(dtype=x.dtype)             # :32  -reduce_mean(log_sigmoid)
    l2 = _l2(u, p, n)                                      # bpr.py:35 (bias NOT included)
    return loss, l2, x


def bpr_grads(U, V, b, uid, pid, nid):
    """Per-occurrence gradients of J = loss + l2_loss (the tuple target of
    tf2_examples/bpr_citeulike.py:36-37 is summed by GradientTape) w.r.t. the
    gathered slices, all evaluated on the PRE-step tables.
    Returns dict(gu, gp, gn [B,D]; gbp, gbn [B])."""
    u, p, n = U[uid], V[pid], V[nid]
    _, _, x = bpr_forward(U, V, b, uid, pid, nid)
    B = x.shape[0]
    dt = x.dtype
    # d loss / d x_k = -(1/B) * sigmoid(-x) * [x >= -30]   (Maximum routes the
    # gradient to its first argument on >=)
    g = (-_sigmoid(-x) * (x >= -30.0) / dt.type(B)).astype(dt)
    gu = g[:, None] * (p - n) + u
    gp = g[:, None] * u + p
    gn = -g[:, None] * u + n
    return dict(gu=gu, gp=gp, gn=gn, gbp=g, gbn=-g, g=g)


def bpr_inference(U, V, b, uid):
    """bpr.py:39-43:  U[uid] @ V.T + b."""
    return U[uid] @ V.T + b[:, 0][None, :]


# --------------------------------------------------------------------------
# UCML  (openrec/tf2/recommenders/ucml.py:21-42)
# --------------------------------------------------------------------------
def ucml_forward(U, V, b, uid, pid, nid, margin=0.5):
    u, p, n = U[uid], V[pid], V[nid]                       # ucml.py:23-27
    dpos = ((u - p) ** 2).sum(1)                           # :29-31
    dneg = ((u - n) ** 2).sum(1)                           # :32-34
    diff = (-dpos + b[pid, 0]) - (-dneg + b[nid, 0])       # :35-37
    h = np.asarray(margin, diff.dtype) - diff
    loss = np.maximum(h, 0).sum(dtype=diff.dtype)          # :39  reduce_SUM
    l2 = _l2(u, p, n)                                      # :40
    return loss, l2, h


def ucml_grads(U, V, b, uid, pid, nid, margin=0.5):
    u, p, n = U[uid], V[pid], V[nid]
    _, _, h = ucml_forward(U, V, b, uid, pid, nid, margin)
    a = (h >= 0).astype(h.dtype)                           # Maximum: ties go to arg 0
    gu = -2 * a[:, None] * (p - n) + u
    gp = -2 * a[:, None] * (u - p) + p
    gn = 2 * a[:, None] * (u - n) + n
    return dict(gu=gu, gp=gp, gn=gn, gbp=-a, gbn=a, g=a)


def ucml_inference(U, V, b, uid):
    """ucml.py:50-53."""
    u = U[uid]
    return -((u[:, None, :] - V[None, :, :]) ** 2).sum(-1) + b[:, 0][None, :]


def censor(W, ids, min_norm=0.1):
    """LatentFactor.censor (latent_factor.py:17-23): for first-occurrence
    unique ids,  W[i] <- W[i] / max(||W[i]||_2, 0.1).  In place."""
    uid = tf_unique(ids)
    g = W[uid]
    norm = np.sqrt((g * g).sum(1, keepdims=True, dtype=W.dtype))
    W[uid] = g / np.maximum(norm, np.asarray(min_norm, W.dtype))
    return uid


# --------------------------------------------------------------------------
# GMF / WRMF (pointwise)  gmf.py:22-34, wrmf.py:21-34, pointwise_mse_loss.py:18-31
# --------------------------------------------------------------------------
def gmf_forward(U, V, b, w, uid, iid, label):
    """w: Dense(1, use_bias=False) kernel, shape [D, 1] (gmf.py:19)."""
    u, i = U[uid], V[iid]
    z = (u * i) @ w[:, 0] + b[iid, 0]                      # gmf.py:28
    # Keras BinaryCrossentropy(from_logits=True) -> mean over batch of
    # max(z,0) - z*y + log(1+exp(-|z|))
    y = label.astype(z.dtype)
    per = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    loss = per.mean(dtype=z.dtype)
    l2 = _l2(u, i) + _l2(w)                                # gmf.py:31-32
    return loss, l2, z


def gmf_grads(U, V, b, w, uid, iid, label):
    u, i = U[uid], V[iid]
    _, _, z = gmf_forward(U, V, b, w, uid, iid, label)
    B = z.shape[0]
    y = label.astype(z.dtype)
    gz = ((_sigmoid(z) - y) / z.dtype.type(B)).astype(z.dtype)
    gu = gz[:, None] * (i * w[:, 0][None, :]) + u
    gi = gz[:, None] * (u * w[:, 0][None, :]) + i
    gw = ((u * i) * gz[:, None]).sum(0, dtype=z.dtype)[:, None] + w
    return dict(gu=gu, gi=gi, gb=gz, gw=gw, g=gz)


def wrmf_forward(U, V, b, uid, iid, label, a=1.0, b_w=1.0, sigmoid=False):
    u, i = U[uid], V[iid]
    pred = (u * i).sum(1) + b[iid, 0]                      # pointwise_mse_loss.py:22-28
    if sigmoid:
        pred = _sigmoid(pred)
    y = label.astype(pred.dtype)
    c = (pred.dtype.type(a) - pred.dtype.type(b_w)) * y + pred.dtype.type(b_w)   # :30
    loss = (c * (y - pred) ** 2).sum(dtype=pred.dtype)     # :31 reduce_SUM
    l2 = _l2(u, i)                                         # wrmf.py:32
    return loss, l2, pred


def wrmf_grads(U, V, b, uid, iid, label, a=1.0, b_w=1.0, sigmoid=False):
    u, i = U[uid], V[iid]
    _, _, pred = wrmf_forward(U, V, b, uid, iid, label, a, b_w, sigmoid)
    y = label.astype(pred.dtype)
    c = (pred.dtype.type(a) - pred.dtype.type(b_w)) * y + pred.dtype.type(b_w)
    gs = -2 * c * (y - pred)
    if sigmoid:
        gs = gs * pred * (1 - pred)
    gu = gs[:, None] * i + u
    gi = gs[:, None] * u + i
    return dict(gu=gu, gi=gi, gb=gs, g=gs)


# --------------------------------------------------------------------------
# Keras OptimizerV2 sparse-apply rules (TF 2.0.x)
# --------------------------------------------------------------------------
def _dedup_sum(idx, vals):
    """OptimizerV2._deduplicate_indexed_slices: unique + unsorted_segment_sum."""
    uniq, inv = np.unique(np.asarray(idx), return_inverse=True)
    out = np.zeros((uniq.shape[0],) + vals.shape[1:], vals.dtype)
    np.add.at(out, inv, vals)
    return uniq, out


class SGD:
    """keras.optimizers.SGD(momentum=0): ``var.scatter_add(idx, -lr*grad)``;
    every occurrence accumulated, no dedup."""
    kind = "sgd"

    def __init__(self, lr=0.01):
        self.lr = lr

    def apply(self, var, idx, grad, key=None):
        np.add.at(var, np.asarray(idx), (-var.dtype.type(self.lr)) * grad)

    def apply_dense(self, var, grad, key=None):
        var -= var.dtype.type(self.lr) * grad


class Adagrad:
    """keras.optimizers.Adagrad: dedup-sum G, acc += G^2,
    var -= lr*G/(sqrt(acc)+eps)."""
    kind = "adagrad"

    def __init__(self, lr=0.001, initial_accumulator_value=0.1, epsilon=1e-7):
        self.lr, self.init_acc, self.eps = lr, initial_accumulator_value, epsilon
        self.acc = {}

    def _acc(self, var, key):
        key = id(var) if key is None else key
        if key not in self.acc:
            self.acc[key] = np.full_like(var, self.init_acc)
        return self.acc[key]

    def apply(self, var, idx, grad, key=None):
        acc = self._acc(var, key)
        uniq, G = _dedup_sum(idx, grad)
        acc[uniq] += G * G
        var[uniq] -= var.dtype.type(self.lr) * G / (np.sqrt(acc[uniq]) + var.dtype.type(self.eps))

    def apply_dense(self, var, grad, key=None):
        acc = self._acc(var, key)
        acc += grad * grad
        var -= var.dtype.type(self.lr) * grad / (np.sqrt(acc) + var.dtype.type(self.eps))


class AdamTFSparse:
    """keras.optimizers.Adam._resource_apply_sparse in TF 2.0.x ("dense decay"):
    m <- b1*m (whole table); m[idx] += (1-b1)*G; v likewise; then the update
    var -= lr_t * m/(sqrt(v)+eps) sweeps the WHOLE table
    (tf2_examples/bpr_citeulike.py:31 uses Adam with defaults)."""
    kind = "adam"

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.m, self.v, self.t = {}, {}, 0

    def begin_step(self):
        self.t += 1

    def _slots(self, var, key):
        key = id(var) if key is None else key
        if key not in self.m:
            self.m[key] = np.zeros_like(var)
            self.v[key] = np.zeros_like(var)
        return self.m[key], self.v[key]

    def _lr_t(self, dt):
        t = self.t
        return dt.type(self.lr * np.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t))

    def apply(self, var, idx, grad, key=None):
        m, v = self._slots(var, key)
        dt = var.dtype
        uniq, G = _dedup_sum(idx, grad)
        m *= dt.type(self.b1)
        m[uniq] += dt.type(1 - self.b1) * G
        v *= dt.type(self.b2)
        v[uniq] += dt.type(1 - self.b2) * G * G
        var -= self._lr_t(dt) * m / (np.sqrt(v) + dt.type(self.eps))

    def apply_dense(self, var, grad, key=None):
        m, v = self._slots(var, key)
        dt = var.dtype
        m *= dt.type(self.b1); m += dt.type(1 - self.b1) * grad
        v *= dt.type(self.b2); v += dt.type(1 - self.b2) * grad * grad
        var -= self._lr_t(dt) * m / (np.sqrt(v) + dt.type(self.eps))


# --------------------------------------------------------------------------
# whole train steps  (tf2_examples/bpr_citeulike.py:33-39)
# --------------------------------------------------------------------------
def _pair_apply(opt, U, V, b, uid, pid, nid, gr, keys=("U", "V", "b")):
    # `keys` name the optimizer's slot variables: one optimizer shared by several models (apply_gradients only touches
    # the variables it is handed, tf2_examples/bpr_citeulike.py:38) keeps one (m, v) pair per VARIABLE
    if hasattr(opt, "begin_step"):
        opt.begin_step()
    opt.apply(U, uid, gr["gu"], key=keys[0])
    # item IndexedSlices = concat of the two lookups of the same variable
    opt.apply(V, np.concatenate([pid, nid]), np.concatenate([gr["gp"], gr["gn"]]), key=keys[1])
    opt.apply(b, np.concatenate([pid, nid]),
              np.concatenate([gr["gbp"], gr["gbn"]])[:, None], key=keys[2])


def bpr_step(U, V, b, uid, pid, nid, opt, keys=("U", "V", "b")):
    """One ``train_step``: forward on pre-step tables, gradients of
    loss + l2_loss, optimizer sparse apply.  Returns (loss, l2_loss)."""
    loss, l2, _ = bpr_forward(U, V, b, uid, pid, nid)
    gr = bpr_grads(U, V, b, uid, pid, nid)
    _pair_apply(opt, U, V, b, uid, pid, nid, gr, keys)
    return loss, l2


def ucml_step(U, V, b, uid, pid, nid, opt, margin=0.5, do_censor=True, min_norm=0.1):
    """UCML.call + apply + (optionally) censor_vec (ucml.py:44-48: users, then
    p items, then n items -- sequential on the same item table)."""
    loss, l2, _ = ucml_forward(U, V, b, uid, pid, nid, margin)
    gr = ucml_grads(U, V, b, uid, pid, nid, margin)
    _pair_apply(opt, U, V, b, uid, pid, nid, gr)
    if do_censor:
        censor(U, uid, min_norm)
        censor(V, pid, min_norm)
        censor(V, nid, min_norm)
    return loss, l2


def gmf_step(U, V, b, w, uid, iid, label, opt):
    loss, l2, _ = gmf_forward(U, V, b, w, uid, iid, label)
    gr = gmf_grads(U, V, b, w, uid, iid, label)
    if hasattr(opt, "begin_step"):
        opt.begin_step()
    opt.apply(U, uid, gr["gu"], key="U")
    opt.apply(V, iid, gr["gi"], key="V")
    opt.apply(b, iid, gr["gb"][:, None], key="b")
    opt.apply_dense(w, gr["gw"], key="w")
    return loss, l2


def wrmf_step(U, V, b, uid, iid, label, opt, a=1.0, b_w=1.0, sigmoid=False):
    loss, l2, _ = wrmf_forward(U, V, b, uid, iid, label, a, b_w, sigmoid)
    gr = wrmf_grads(U, V, b, uid, iid, label, a, b_w, sigmoid)
    if hasattr(opt, "begin_step"):
        opt.begin_step()
    opt.apply(U, uid, gr["gu"], key="U")
    opt.apply(V, iid, gr["gi"], key="V")
    opt.apply(b, iid, gr["gb"][:, None], key="b")
    return loss, l2
