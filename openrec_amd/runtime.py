"""Thin object layer over the C ABI: Context (device + stream), Table (fp32
[rows, dim] in HBM), Optimizer (Keras sparse-apply rules), and the fused train
steps.  Host-side plumbing only; all arithmetic runs in libopenrec_hip.so."""
from __future__ import annotations

import ctypes
import weakref
from ctypes import byref, c_double, c_int64, c_void_p

import os

import numpy as np

from . import _ffi
from ._ffi import check

_default_ctx = None


def _is_device_tensor(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


class DevicePtr:
    """A raw device pointer + element count (ids already resident in HBM)."""

    def __init__(self, ptr, n, keepalive=None):
        self.ptr, self.n, self.keepalive = int(ptr), int(n), keepalive


def _ids_arg(x):
    """-> (pointer, n, on_device, keepalive)"""
    if isinstance(x, DevicePtr):
        return x.ptr, x.n, True, x
    if _is_device_tensor(x):
        assert str(x.dtype).endswith("int32"), "device ids must be int32"
        assert x.is_contiguous()
        return x.data_ptr(), x.numel(), True, x
    if hasattr(x, "numpy") and not isinstance(x, np.ndarray):      # lazy / cpu tensors
        x = x.numpy()
    a = np.ascontiguousarray(x, dtype=np.int32).reshape(-1)       # tf.cast(ids, int32) in Embedding.call
    return a.ctypes.data, a.size, False, a


class Context:
    def __init__(self, device=0, stream=None):
        self._lib = _ffi.load()
        h = c_void_p()
        check(self._lib.orx_ctx_create(int(device), c_void_p(stream) if stream else None, byref(h)))
        self._h = h
        self.device = int(device)
        self._fin = weakref.finalize(self, self._lib.orx_ctx_destroy, h)

    def synchronize(self):
        check(self._lib.orx_synchronize(self._h))

    def wait_stream(self, stream_handle):
        """later calls on this context wait for the work `stream_handle` (a hipStream_t as int; 0 = default stream) holds now"""
        check(self._lib.orx_ctx_wait_stream(self._h, c_void_p(int(stream_handle)) if stream_handle else None))

    def after_torch(self, *tensors):
        """device tensors produced by torch ops are read by the library on ITS stream: order it behind torch's current one"""
        for t in tensors:
            if _is_device_tensor(t):
                import torch
                self.wait_stream(torch.cuda.current_stream(t.device).cuda_stream)
                return

    def check_index_error(self):
        check(self._lib.orx_check_index_error(self._h))

    # ---- kernel-time sampling ------------------------------------------
    def prof_enable(self, on=True):
        check(self._lib.orx_prof_enable(self._h, 1 if on else 0))

    def prof_reset(self):
        check(self._lib.orx_prof_reset(self._h))

    def stat(self, what):
        """orx_ctx_stat: 'pairs' | 'max_dup' | 'nowait_calls' | 'quiet' of the most recent exact pairwise call's plan"""
        v = c_int64()
        check(self._lib.orx_ctx_stat(self._h, {"pairs": 0, "max_dup": 1, "nowait_calls": 2, "quiet": 3}[what], byref(v)))
        return int(v.value)

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        """orx_copy_bandwidth: GB/s (read + write) of a float4 streaming copy over two buffers of `nbytes` each"""
        v = c_double()
        check(self._lib.orx_copy_bandwidth(self._h, int(nbytes), int(reps), byref(v)))
        return float(v.value)

    def prof_get(self):
        out = {}
        for kid, name in _ffi.KERNEL_NAMES.items():
            ms, n = c_double(), c_int64()
            check(self._lib.orx_prof_get(self._h, kid, byref(ms), byref(n)))
            out[name] = dict(total_ms=ms.value, launches=n.value)
        return out


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Table:
    """fp32 [rows, dim] embedding table resident in HBM."""

    def __init__(self, rows, dim, ctx=None, device_ptr=None, keepalive=None):
        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        h = c_void_p()
        if device_ptr is None:
            check(self._lib.orx_table_create(self.ctx._h, int(rows), int(dim), byref(h)))
        else:
            check(self._lib.orx_table_wrap(self.ctx._h, c_void_p(int(device_ptr)), int(rows), int(dim), byref(h)))
        self._h = h
        self.rows, self.dim = int(rows), int(dim)
        self._keepalive = (keepalive, self.ctx)
        self._fin = weakref.finalize(self, self._lib.orx_table_destroy, h)
        self.pre_access = None          # callable run before any host-visible access (queued train steps flush here)

    pre_access = None

    def _sync_pending(self):
        if self.pre_access is not None:
            self.pre_access()

    @property
    def shape(self):
        return (self.rows, self.dim)

    @property
    def device_ptr(self):
        return self._lib.orx_table_device_ptr(self._h)

    def init_uniform(self, lo=-0.05, hi=0.05, seed=0):
        self._sync_pending()
        check(self._lib.orx_table_init_uniform(self._h, lo, hi, int(seed) & (2 ** 64 - 1)))
        return self

    def fill(self, v):
        self._sync_pending()
        check(self._lib.orx_table_fill(self._h, float(v)))
        return self

    def read(self, row0=0, nrows=None):
        self._sync_pending()
        nrows = self.rows - row0 if nrows is None else nrows
        out = np.empty((nrows, self.dim), np.float32)
        check(self._lib.orx_table_read(self._h, int(row0), int(nrows), out.ctypes.data))
        return out

    def numpy(self):
        return self.read()

    def write(self, values, row0=0):
        self._sync_pending()
        a = np.ascontiguousarray(values, np.float32).reshape(-1, self.dim)
        check(self._lib.orx_table_write(self._h, int(row0), a.shape[0], a.ctypes.data))
        return self

    def gather(self, ids):
        self._sync_pending()
        ptr, n, dev, keep = _ids_arg(ids)
        if dev:
            raise ValueError("Table.gather returns host rows; pass host ids (use gather_rows for device buffers)")
        out = np.empty((n, self.dim), np.float32)
        check(self._lib.orx_table_gather(self._h, ptr, n, out.ctypes.data, 0))
        return out

    def censor(self, ids, min_norm=0.1):
        self._sync_pending()
        ptr, n, dev, keep = _ids_arg(ids)
        check(self._lib.orx_table_censor(self._h, ptr, n, float(min_norm), _ffi.ORX_IDS_DEVICE if dev else 0))


class Optimizer:
    KINDS = {"sgd": _ffi.ORX_SGD, "adagrad": _ffi.ORX_ADAGRAD, "adam": _ffi.ORX_ADAM}

    def __init__(self, kind, lr, p0=0.0, p1=0.0, p2=0.0, ctx=None):
        self.ctx = ctx or default_context()
        self._lib = self.ctx._lib
        self.kind = kind
        h = c_void_p()
        check(self._lib.orx_opt_create(self.ctx._h, self.KINDS[kind], lr, p0, p1, p2, byref(h)))
        self._h = h
        self._tables = []          # keep tables alive while the optimizer holds slots for them
        self._fin = weakref.finalize(self, self._lib.orx_opt_destroy, h)

    @classmethod
    def sgd(cls, lr=0.01, ctx=None):
        return cls("sgd", lr, ctx=ctx)

    @classmethod
    def adagrad(cls, lr=0.001, initial_accumulator_value=0.1, epsilon=1e-7, ctx=None):
        return cls("adagrad", lr, initial_accumulator_value, epsilon, ctx=ctx)

    @classmethod
    def adam(cls, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, ctx=None):
        return cls("adam", lr, beta_1, beta_2, epsilon, ctx=ctx)

    def set_lr(self, lr):
        check(self._lib.orx_opt_set_lr(self._h, lr))

    @property
    def step(self):
        """the step counter (Keras `optimizer.iterations`)"""
        t = c_int64(0)
        check(self._lib.orx_opt_get_step(self._h, byref(t)))
        return int(t.value)

    @step.setter
    def step(self, value):
        check(self._lib.orx_opt_set_step(self._h, int(value)))

    def advance(self, tables=None):
        """One optimizer step begins (Keras `iterations` += 1) for a host that applies gradients through `apply_rows`.
        `tables`: the tables this step updates -- any other table lazily applied under this optimizer is finished first
        and takes no Adam decay for the step (Keras updates only the variables handed to apply_gradients); None: the
        optimizer's own tables, all of them."""
        if tables is None:
            check(self._lib.orx_opt_advance(self._h, None, -1))
        else:
            arr = (c_void_p * len(tables))(*[t._h for t in tables])
            check(self._lib.orx_opt_advance(self._h, ctypes.cast(arr, c_void_p), len(tables)))

    def slot(self, table, slot=0):
        out = np.empty((table.rows, table.dim), np.float32)
        check(self._lib.orx_opt_slot_read(self._h, table._h, slot, 0, table.rows, out.ctypes.data))
        return out

    def set_slot(self, table, values, slot=0):
        a = np.ascontiguousarray(values, np.float32).reshape(table.rows, table.dim)
        check(self._lib.orx_opt_slot_write(self._h, table._h, slot, 0, table.rows, a.ctypes.data))

    def slot_rows(self, table, slot, row0, nrows):
        """rows [row0, row0 + nrows) of an optimizer slot of `table` (checkpoints stream slots in row ranges)"""
        out = np.empty((nrows, table.dim), np.float32)
        check(self._lib.orx_opt_slot_read(self._h, table._h, int(slot), int(row0), int(nrows), out.ctypes.data))
        return out

    def set_slot_rows(self, table, values, slot, row0):
        a = np.ascontiguousarray(values, np.float32).reshape(-1, table.dim)
        check(self._lib.orx_opt_slot_write(self._h, table._h, int(slot), int(row0), a.shape[0], a.ctypes.data))


def pairwise_step(model, opt, user, item, bias, uid, pid, nid, K=1, B=None, id_stride=None,
                  margin=0.5, hogwild=False, no_l2=False, want_loss=True, censor=False):
    """K fused train steps.  Returns (loss[K], l2[K]) as numpy float32 when
    want_loss, else None (fully asynchronous)."""
    lib = user.ctx._lib
    pu, nu, du, k0 = _ids_arg(uid)
    pp, npn, dp, k1 = _ids_arg(pid)
    pn, nn, dn, k2 = _ids_arg(nid)
    assert du == dp == dn, "ids must be all host or all device"
    assert nu == npn == nn, "id arrays differ in length"
    user.ctx.after_torch(uid, pid, nid)
    if B is None:
        B = nu // K
    if id_stride is None:
        id_stride = B
    flags = ((_ffi.ORX_IDS_DEVICE if du else 0) | (_ffi.ORX_HOGWILD if hogwild else 0)
             | (_ffi.ORX_NO_L2 if no_l2 else 0) | (_ffi.ORX_CENSOR if censor else 0))
    mid = {"bpr": _ffi.ORX_BPR, "ucml": _ffi.ORX_UCML}[model]
    if want_loss:
        loss = np.empty(K, np.float32)
        l2 = np.empty(K, np.float32)
        lp, l2p = loss.ctypes.data, l2.ctypes.data
    else:
        loss = l2 = None
        lp = l2p = None
    check(lib.orx_pairwise_step(user.ctx._h, mid, opt._h, user._h, item._h, bias._h, pu, pp, pn,
                                int(K), int(B), int(id_stride), float(margin), flags, lp, l2p))
    opt._tables = list({id(t): t for t in (opt._tables + [user, item, bias])}.values())
    return (loss, l2) if want_loss else None


def pairwise_reserve(opt, user, item, bias, K, B):
    """Pre-size every per-call device buffer for calls of up to K steps of B triplets."""
    check(user.ctx._lib.orx_pairwise_reserve(user.ctx._h, opt._h, user._h, item._h, bias._h, int(K), int(B)))
    opt._tables = list({id(t): t for t in (opt._tables + [user, item, bias])}.values())


def pairwise_loss(model, user, item, bias, uid, pid, nid, margin=0.5):
    lib = user.ctx._lib
    pu, nu, du, k0 = _ids_arg(uid)
    pp, _, dp, k1 = _ids_arg(pid)
    pn, _, dn, k2 = _ids_arg(nid)
    mid = {"bpr": _ffi.ORX_BPR, "ucml": _ffi.ORX_UCML}[model]
    loss = np.empty(1, np.float32)
    l2 = np.empty(1, np.float32)
    check(lib.orx_pairwise_loss(user.ctx._h, mid, user._h, item._h, bias._h, pu, pp, pn, nu, float(margin),
                                _ffi.ORX_IDS_DEVICE if du else 0, loss.ctypes.data, l2.ctypes.data))
    return float(loss[0]), float(l2[0])


def _label_arg(x):
    if _is_device_tensor(x):
        assert str(x.dtype).endswith("float32") and x.is_contiguous()
        return x.data_ptr(), x.numel(), True, x
    if hasattr(x, "numpy") and not isinstance(x, np.ndarray):
        x = x.numpy()
    a = np.ascontiguousarray(x, dtype=np.float32).reshape(-1)
    return a.ctypes.data, a.size, False, a


_POINT = {"gmf": _ffi.ORX_GMF, "wrmf": _ffi.ORX_WRMF}


def pointwise_step(model, opt, user, item, bias, w, uid, iid, label, K=1, B=None, id_stride=None,
                   a=1.0, b_w=1.0, hogwild=False, no_l2=False, want_loss=True, sigmoid=False):
    """K fused GMF / WRMF train steps (gmf.py:22-34, wrmf.py:21-34); sigmoid: PointwiseMSELoss(sigmoid=True)
    (pointwise_mse_loss.py:24-25; WRMF only)."""
    lib = user.ctx._lib
    pu, nu, du, k0 = _ids_arg(uid)
    pi, ni, di, k1 = _ids_arg(iid)
    pl, nl, dl, k2 = _label_arg(label)
    assert du == di == dl and nu == ni == nl
    user.ctx.after_torch(uid, iid, label)
    if B is None:
        B = nu // K
    if id_stride is None:
        id_stride = B
    flags = (_ffi.ORX_IDS_DEVICE if du else 0) | (_ffi.ORX_HOGWILD if hogwild else 0) | (_ffi.ORX_NO_L2 if no_l2 else 0) \
        | (_ffi.ORX_POINT_SIGMOID if sigmoid else 0)
    loss = np.empty(K, np.float32) if want_loss else None
    l2 = np.empty(K, np.float32) if want_loss else None
    check(lib.orx_pointwise_step(user.ctx._h, _POINT[model], opt._h, user._h, item._h, bias._h,
                                 w._h if w is not None else None, pu, pi, pl, int(K), int(B), int(id_stride),
                                 float(a), float(b_w), flags,
                                 loss.ctypes.data if want_loss else None, l2.ctypes.data if want_loss else None))
    opt._tables = list({id(t): t for t in (opt._tables + [user, item, bias] + ([w] if w is not None else []))}.values())
    return (loss, l2) if want_loss else None


def pointwise_loss(model, user, item, bias, w, uid, iid, label, a=1.0, b_w=1.0, sigmoid=False):
    lib = user.ctx._lib
    pu, nu, du, k0 = _ids_arg(uid)
    pi, _, di, k1 = _ids_arg(iid)
    pl, _, dl, k2 = _label_arg(label)
    loss = np.empty(1, np.float32)
    l2 = np.empty(1, np.float32)
    check(lib.orx_pointwise_loss(user.ctx._h, _POINT[model], user._h, item._h, bias._h,
                                 w._h if w is not None else None, pu, pi, pl, nu, float(a), float(b_w),
                                 (_ffi.ORX_IDS_DEVICE if du else 0) | (_ffi.ORX_POINT_SIGMOID if sigmoid else 0), loss.ctypes.data, l2.ctypes.data))
    return float(loss[0]), float(l2[0])


class _HostView(np.lib.mixins.NDArrayOperatorsMixin):
    """An array kept in another form (device memory, item lists) that turns into its NumPy value when someone looks:
    `np.asarray(x)`, `x.numpy()`, indexing, arithmetic and any ndarray attribute all go through `_dense()`."""
    _host = None

    def numpy(self):
        if self._host is None:
            self._host = self._dense()
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        return getattr(ufunc, method)(*[x.numpy() if isinstance(x, _HostView) else x for x in inputs], **kw)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        return self.numpy()[i]

    def __iter__(self):
        return iter(self.numpy())

    def __getattr__(self, name):                       # (only reached for what the class does not define)
        if name.startswith("__"):
            raise AttributeError(name)
        return getattr(self.numpy(), name)

    ndim = 2


class DeviceScores(_HostView):
    """[n, items] fp32 scores of `Recommender.inference` held in device memory (a torch tensor).  The metrics read them there;
    the host copy (4 bytes x n x items over PCIe) is made only if the script looks at the values."""
    dtype = np.dtype(np.float32)

    def __init__(self, ctx, tensor):
        self.ctx, self.tensor, self.shape = ctx, tensor, tuple(tensor.shape)

    def _dense(self):
        return self.tensor.cpu().numpy()


class SparseMask(_HostView):
    """A batch of boolean item masks [n, items] as one sorted list of distinct items per row (CSR): what
    `Dataset.evaluation` hands out instead of the dense rows of openrec/tf2/data/dataset.py:60-82.  Dense on demand."""
    dtype = np.dtype(bool)

    def __init__(self, ptr, items, n_items):
        self.ptr = np.ascontiguousarray(ptr, np.int64); self.items = np.ascontiguousarray(items, np.int32)
        self.shape = (self.ptr.size - 1, int(n_items))

    @classmethod
    def from_lists(cls, lists, n_items):
        rows = [np.unique(np.asarray(r, np.int64)) for r in lists]
        for r in rows:
            if r.size and (r[0] < 0 or r[-1] >= n_items):
                raise IndexError(f"item id outside [0, {n_items})")
        ptr = np.zeros(len(rows) + 1, np.int64); np.cumsum([r.size for r in rows], out=ptr[1:])
        return cls(ptr, np.concatenate(rows).astype(np.int32) if rows else np.zeros(0, np.int32), n_items)

    @classmethod
    def from_dense(cls, mask):
        m = np.asarray(mask).astype(bool)
        r, c = np.nonzero(m)
        ptr = np.zeros(m.shape[0] + 1, np.int64); np.cumsum(np.bincount(r, minlength=m.shape[0]), out=ptr[1:])
        return cls(ptr, c.astype(np.int32), m.shape[1])

    def row(self, q):
        return self.items[self.ptr[q]:self.ptr[q + 1]]

    def _dense(self):
        m = np.zeros(self.shape, bool)
        m[np.repeat(np.arange(self.shape[0]), np.diff(self.ptr)), self.items] = True
        return m


def score_all_items(kind, user, item, bias, uid, w=None, device=False):
    """Recommender.inference: scores of the given users against ALL items -> [n, item_rows] (host array, or with
    `device=True` a DeviceScores that stays in HBM until someone reads it)."""
    lib = user.ctx._lib
    ptr, n, dev, keep = _ids_arg(uid)
    if dev:
        raise ValueError("score_all_items takes host ids")
    k = {"dot": 0, "l2": 1, "gmf": 2}[kind]
    if device:
        try:
            import torch
        except ImportError:                  # torch is optional on the single-GPU path: the scores then come back as a host array
            torch = None
    if device and torch is not None:
        out = torch.empty((n, item.rows), dtype=torch.float32, device=torch.device("cuda", user.ctx.device))
        user.ctx.after_torch(out)            # (a cached block may still have work of torch's stream pending: the library's stream waits for it)
        check(lib.orx_score_all_items_device(user.ctx._h, k, user._h, item._h, bias._h, w._h if w is not None else None,
                                             ptr, n, out.data_ptr()))
        return DeviceScores(user.ctx, out)
    out = np.empty((n, item.rows), np.float32)
    check(lib.orx_score_all_items(user.ctx._h, k, user._h, item._h, bias._h, w._h if w is not None else None,
                                  ptr, n, out.ctypes.data))
    return out


class _BorrowedTable(Table):
    """A table handle owned by another object (e.g. a DLRM model's parameter)."""

    def __init__(self, ctx, handle, owner):
        self.ctx, self._lib, self._h = ctx, ctx._lib, handle
        self.rows, self.dim = int(self._lib.orx_table_rows(handle)), int(self._lib.orx_table_dim(handle))
        self._keepalive = (owner, ctx)


class DLRMModel:
    """Device-side DLRM (recommenders/dlrm.py:6-100): combined embedding table,
    bottom / top MLPs, feature interaction, loss; `step` = forward + backward +
    optimizer apply of tf2_examples/dlrm_criteo.py:42-48."""

    def __init__(self, m_spa, ln_emb, ln_bot, ln_top, dense_dim, arch_interaction_itself=False, sigmoid_bot=False,
                 sigmoid_top=True, loss_func="mse", loss_threshold=0.0, reference_compat=True, seed=0, ctx=None,
                 fp16_mlp=False, no_emb=False):
        self.ctx = ctx or default_context()
        lib = self._lib = self.ctx._lib
        self.m_spa, self.ln_emb, self.ln_bot, self.ln_top = int(m_spa), [int(x) for x in ln_emb], list(ln_bot), list(ln_top)
        self.dense_dim = int(dense_dim)
        self.loss_func = loss_func
        flags = ((_ffi.ORX_DLRM_INTERACT_ITSELF if arch_interaction_itself else 0)
                 | (_ffi.ORX_DLRM_SIGMOID_BOT if sigmoid_bot else 0) | (_ffi.ORX_DLRM_SIGMOID_TOP if sigmoid_top else 0)
                 | (_ffi.ORX_DLRM_LOSS_BCE if loss_func == "bce" else 0)
                 | (_ffi.ORX_DLRM_REFERENCE_COMPAT if reference_compat else 0)
                 | (_ffi.ORX_DLRM_FP16_MLP if fp16_mlp else 0) | (_ffi.ORX_DLRM_NO_EMB if no_emb else 0))
        if loss_func not in ("mse", "bce"):
            raise ValueError("loss_func=%s is not supported" % loss_func)          # dlrm.py:56-61
        emb = (ctypes.c_int64 * len(self.ln_emb))(*self.ln_emb)
        bot = (ctypes.c_int32 * len(ln_bot))(*ln_bot)
        top = (ctypes.c_int32 * len(ln_top))(*ln_top)
        h = c_void_p()
        check(lib.orx_dlrm_create(self.ctx._h, self.m_spa, len(self.ln_emb), ctypes.cast(emb, c_void_p),
                                  len(ln_bot), ctypes.cast(bot, c_void_p), len(ln_top), ctypes.cast(top, c_void_p),
                                  self.dense_dim, flags, float(loss_threshold), int(seed), byref(h)))
        self._h = h
        self._fin = weakref.finalize(self, lib.orx_dlrm_destroy, h)
        self.offsets = np.concatenate([[0], np.cumsum(self.ln_emb)[:-1]]).astype(np.int64)

    def param(self, kind, layer=0):
        k = {"emb": 0, "bot_w": 1, "bot_b": 2, "top_w": 3, "top_b": 4}[kind]
        h = c_void_p()
        check(self._lib.orx_dlrm_param(self._h, k, int(layer), byref(h)))
        return _BorrowedTable(self.ctx, h, self)

    def emb_table(self, f):
        """numpy view helper: rows of embedding table f inside the combined table"""
        return int(self.offsets[f]), self.ln_emb[f]

    @staticmethod
    def _host(x, dtype):
        if hasattr(x, "numpy") and not isinstance(x, np.ndarray):
            x = x.numpy()
        return np.ascontiguousarray(x, dtype=dtype)

    def step_device(self, opt, dense_ptr, sparse_ptr, label_ptr, K, B, want_loss=False):
        """K steps on batches that already sit in HBM (device pointers, layout as `step`)."""
        loss = np.empty(K, np.float32) if want_loss else None
        check(self._lib.orx_dlrm_step(self._h, opt._h, dense_ptr, sparse_ptr, label_ptr, int(K), int(B), _ffi.ORX_IDS_DEVICE,
                                      loss.ctypes.data if want_loss else None))
        opt._tables.append(self)
        return loss

    def step(self, opt, dense, sparse, label, K=1, want_loss=True):
        d, s, y = self._host(dense, np.float32), self._host(sparse, np.int32), self._host(label, np.float32)
        B = y.size // K
        assert d.size == K * B * self.dense_dim and s.size == K * B * len(self.ln_emb)
        loss = np.empty(K, np.float32) if want_loss else None
        check(self._lib.orx_dlrm_step(self._h, opt._h, d.ctypes.data, s.ctypes.data, y.ctypes.data, K, B, 0,
                                      loss.ctypes.data if want_loss else None))
        opt._tables.append(self)
        return loss

    # ---- hybrid-parallel building blocks (device pointers; openrec_amd/sharded_dlrm.py) ----
    def grads(self, dense_ptr, emb_rows_ptr, label_ptr, B, global_B, emb_grads_ptr, loss_accum_ptr):
        check(self._lib.orx_dlrm_grads(self._h, dense_ptr, emb_rows_ptr, label_ptr, int(B), int(global_B),
                                       emb_grads_ptr, loss_accum_ptr))

    def dense_count(self):
        n = ctypes.c_int64()
        check(self._lib.orx_dlrm_dense_count(self._h, byref(n)))
        return int(n.value)

    def dense_pack(self, flat_ptr):
        check(self._lib.orx_dlrm_dense_pack(self._h, flat_ptr))

    def dense_apply(self, opt, flat_ptr):
        check(self._lib.orx_dlrm_dense_apply(self._h, opt._h, flat_ptr))
        opt._tables.append(self)

    def inference(self, dense, sparse):
        d, s = self._host(dense, np.float32), self._host(sparse, np.int32)
        B = d.size // self.dense_dim
        out = np.empty(B, np.float32)
        check(self._lib.orx_dlrm_inference(self._h, d.ctypes.data, s.ctypes.data, B, 0, out.ctypes.data))
        return out


def rank_metrics(pos_mask, excl_mask, at, pred=None, kind=None, user=None, item=None, bias=None, w=None, uid=None, ctx=None):
    """AUC / NDCG@at / Recall@at per user (openrec/tf2/metrics/ranking_metrics.py).  Either `pred`
    (host scores [n, items]) or the tables + user ids (scores computed on the device)."""
    pos = np.ascontiguousarray(pos_mask, np.uint8)
    excl = np.ascontiguousarray(excl_mask, np.uint8)
    n, items = pos.shape
    atv = np.ascontiguousarray(at, np.float32).reshape(-1)
    c = ctx or (user.ctx if user is not None else default_context())
    auc = np.empty(n, np.float32); ndcg = np.empty((n, atv.size), np.float32); rec = np.empty((n, atv.size), np.float32)
    if pred is not None:
        pr = np.ascontiguousarray(pred, np.float32)
        assert pr.shape == (n, items)
        check(c._lib.orx_rank_metrics(c._h, 0, None, None, None, None, None, pr.ctypes.data, pos.ctypes.data, excl.ctypes.data,
                                      n, items, atv.ctypes.data, atv.size, auc.ctypes.data, ndcg.ctypes.data, rec.ctypes.data))
    else:
        ptr, nn, dev, keep = _ids_arg(uid)
        assert nn == n and not dev
        k = {"dot": 0, "l2": 1, "gmf": 2}[kind]
        check(c._lib.orx_rank_metrics(c._h, k, user._h, item._h, bias._h, w._h if w is not None else None, ptr, None,
                                      pos.ctypes.data, excl.ctypes.data, n, items, atv.ctypes.data, atv.size,
                                      auc.ctypes.data, ndcg.ctypes.data, rec.ctypes.data))
    return dict(auc=auc, ndcg=ndcg, recall=rec)


def rank_metrics_csr(pos, excl, at, pred=None, kind=None, user=None, item=None, bias=None, w=None, uid=None, ctx=None):
    """`rank_metrics` with the masks as SparseMask (item lists): nothing of size n x items crosses PCIe when the scores are
    on the device (`pred` a DeviceScores, or None with the tables + user ids)."""
    assert isinstance(pos, SparseMask) and isinstance(excl, SparseMask) and pos.shape == excl.shape
    n, items = pos.shape
    atv = np.ascontiguousarray(at, np.float32).reshape(-1)
    c = ctx or (user.ctx if user is not None else (pred.ctx if isinstance(pred, DeviceScores) else default_context()))
    auc = np.empty(n, np.float32); ndcg = np.empty((n, atv.size), np.float32); rec = np.empty((n, atv.size), np.float32)
    tail = (n, items, pos.ptr.ctypes.data, pos.items.ctypes.data, excl.ptr.ctypes.data, excl.items.ctypes.data,
            atv.ctypes.data, atv.size, auc.ctypes.data, ndcg.ctypes.data, rec.ctypes.data)
    if pred is not None:
        if isinstance(pred, DeviceScores):
            assert pred.shape == (n, items)
            head = (pred.tensor.data_ptr(), 1)
        else:
            pr = np.ascontiguousarray(pred, np.float32)
            assert pr.shape == (n, items)
            head = (pr.ctypes.data, 0)
        check(c._lib.orx_rank_metrics_csr(c._h, 0, None, None, None, None, None, *head, *tail))
    else:
        ptr, nn, dev, keep = _ids_arg(uid)
        assert nn == n and not dev
        k = {"dot": 0, "l2": 1, "gmf": 2}[kind]
        check(c._lib.orx_rank_metrics_csr(c._h, k, user._h, item._h, bias._h, w._h if w is not None else None, ptr, None, 0, *tail))
    return dict(auc=auc, ndcg=ndcg, recall=rec)


CKPT_PIECE_BYTES = 256 << 20          # tables and slots move through the host in pieces of at most this many bytes


def _stream_rows(read, rows, dim, dst):
    """rows [0, rows) through `read(row0, n) -> [n, dim]` into the (memory-mapped) array `dst`, CKPT_PIECE_BYTES at a time"""
    step = max(1, CKPT_PIECE_BYTES // max(1, 4 * dim))
    for r0 in range(0, rows, step):
        n = min(step, rows - r0)
        dst[r0:r0 + n] = read(r0, n)


def save_checkpoint(path, tables, opt=None, shard=None):
    """Tables (and the optimizer's slots for them) -> a checkpoint.  The tf2 reference has no checkpointing (`save_interval` is
    unused, tf2_examples/bpr_citeulike.py:16); tf1 used tf.train.Saver (tf1/recommenders/recommender.py:430-473).
    `tables`: {name: Table}.

    `path` ending in ".npz": ONE file, every tensor whole in host memory on the way (small models, the round-1 format).
    Otherwise `path` is a DIRECTORY: one .npy per tensor -- `table.<name>.npy`, `slot0.<name>.npy`, `slot1.<name>.npy` -- written
    through a memory map in row ranges of at most CKPT_PIECE_BYTES (a 10 M x 128 table never sits in host memory whole), and
    `manifest.json` (shapes, optimizer kind and step counter).  `shard=(rank, world)`: this rank's files get the suffix
    `.rank<r>of<w>` and the manifest records the sharding (row r of the global table = local row r // world of rank r % world,
    SURVEY.md 8(e)): every rank of a row-sharded job saves its own shard into the same directory."""
    import json
    if str(path).endswith(".npz"):
        out = {}
        for name, t in tables.items():
            out["table/" + name] = t.read()
            if opt is not None and opt.kind in ("adagrad", "adam"):
                out["slot0/" + name] = opt.slot(t, 0)
                if opt.kind == "adam":
                    out["slot1/" + name] = opt.slot(t, 1)
        if opt is not None:
            out["opt/kind"] = np.array(opt.kind)
            out["opt/step"] = np.array(opt.step, np.int64)          # Adam's bias correction resumes where it stopped
        np.savez(path, **out)
        return
    os.makedirs(path, exist_ok=True)
    suffix = "" if shard is None else ".rank%dof%d" % (int(shard[0]), int(shard[1]))
    man = dict(format="openrec_amd-ckpt-1", tensors={}, shard=None if shard is None else dict(rank=int(shard[0]), world=int(shard[1])))
    nslots = 0 if opt is None else {"sgd": 0, "adagrad": 1, "adam": 2}[opt.kind]
    for name, t in tables.items():
        if hasattr(t, "_sync_pending"):
            t._sync_pending()
        man["tensors"][name] = dict(rows=t.rows, dim=t.dim, slots=nslots)
        mm = np.lib.format.open_memmap(os.path.join(path, f"table.{name}{suffix}.npy"), mode="w+", dtype=np.float32, shape=(t.rows, t.dim))
        _stream_rows(t.read, t.rows, t.dim, mm)
        mm.flush(); del mm
        for k in range(nslots):
            mm = np.lib.format.open_memmap(os.path.join(path, f"slot{k}.{name}{suffix}.npy"), mode="w+", dtype=np.float32, shape=(t.rows, t.dim))
            _stream_rows(lambda r0, n, k=k: opt.slot_rows(t, k, r0, n), t.rows, t.dim, mm)
            mm.flush(); del mm
    if opt is not None:
        man["opt"] = dict(kind=opt.kind, step=int(opt.step))
    with open(os.path.join(path, f"manifest{suffix}.json"), "w") as f:
        json.dump(man, f, indent=1)


def load_checkpoint(path, tables, opt=None, shard=None):
    """the inverse of save_checkpoint (same `path` / `shard` conventions); shapes and the optimizer kind are checked"""
    import json
    if str(path).endswith(".npz"):
        z = np.load(path)
        for name, t in tables.items():
            t.write(z["table/" + name])
            if opt is not None and ("slot0/" + name) in z:
                opt.set_slot(t, z["slot0/" + name], 0)
                if ("slot1/" + name) in z:
                    opt.set_slot(t, z["slot1/" + name], 1)
        if opt is not None and "opt/step" in z:
            opt.step = int(z["opt/step"])
        return
    suffix = "" if shard is None else ".rank%dof%d" % (int(shard[0]), int(shard[1]))
    with open(os.path.join(path, f"manifest{suffix}.json")) as f:
        man = json.load(f)
    if man.get("format") != "openrec_amd-ckpt-1":
        raise ValueError(f"{path}: not an openrec_amd checkpoint directory")
    want_shard = None if shard is None else dict(rank=int(shard[0]), world=int(shard[1]))
    if man.get("shard") != want_shard:
        raise ValueError(f"{path}: saved with shard={man.get('shard')}, asked for {want_shard}")
    if opt is not None and "opt" in man and man["opt"]["kind"] != opt.kind:
        raise ValueError(f"{path}: saved with a {man['opt']['kind']} optimizer, loading into {opt.kind}")
    for name, t in tables.items():
        info = man["tensors"].get(name)
        if info is None:
            raise KeyError(f"{path}: no tensor named {name!r}")
        if (info["rows"], info["dim"]) != (t.rows, t.dim):
            raise ValueError(f"{path}: {name} is [{info['rows']}, {info['dim']}], the table [{t.rows}, {t.dim}]")
        step = max(1, CKPT_PIECE_BYTES // max(1, 4 * t.dim))
        mm = np.load(os.path.join(path, f"table.{name}{suffix}.npy"), mmap_mode="r")
        for r0 in range(0, t.rows, step):
            t.write(np.ascontiguousarray(mm[r0:r0 + step]), r0)
        del mm
        if opt is not None:
            for k in range(min(info["slots"], {"sgd": 0, "adagrad": 1, "adam": 2}[opt.kind])):
                mm = np.load(os.path.join(path, f"slot{k}.{name}{suffix}.npy"), mmap_mode="r")
                for r0 in range(0, t.rows, step):
                    opt.set_slot_rows(t, np.ascontiguousarray(mm[r0:r0 + step]), k, r0)
                del mm
    if opt is not None and "opt" in man:
        opt.step = int(man["opt"]["step"])


class DeviceSampler:
    """On-device pairwise sampler over an interaction set (see kernels_sampler.hip).  `raw_data`: the
    structured array the reference's Dataset takes ('user_id', 'item_id')."""

    def __init__(self, raw_data, total_users, total_items, ctx=None):
        self.ctx = ctx or default_context()
        lib = self._lib = self.ctx._lib
        u = np.ascontiguousarray(raw_data["user_id"], np.int32)
        i = np.ascontiguousarray(raw_data["item_id"], np.int32)
        key = np.unique(u.astype(np.int64) * int(total_items) + i)            # sorted by (user, item), distinct
        cu, ci = (key // int(total_items)).astype(np.int64), (key % int(total_items)).astype(np.int32)
        ptr = np.zeros(int(total_users) + 1, np.int64)
        np.add.at(ptr, cu + 1, 1)
        ptr = np.ascontiguousarray(np.cumsum(ptr), np.int64)
        ci = np.ascontiguousarray(ci)
        h = c_void_p()
        check(lib.orx_sampler_create(self.ctx._h, u.ctypes.data, i.ctypes.data, u.size, ptr.ctypes.data, ci.ctypes.data,
                                     int(total_users), int(total_items), byref(h)))
        self._h, self.n_records = h, int(u.size)
        self._fin = weakref.finalize(self, lib.orx_sampler_destroy, h)

    def pairwise(self, seed, first, n, uid, pid, nid):
        """Fill the DEVICE int32 buffers uid / pid / nid (torch tensors or DevicePtr) with samples
        [first, first + n) of stream `seed`."""
        pu, nu, du, _ = _ids_arg(uid); pp, _, dp, _ = _ids_arg(pid); pn, _, dn, _ = _ids_arg(nid)
        assert du and dp and dn and nu >= n, "the sampler writes device buffers"
        check(self._lib.orx_sampler_pairwise(self._h, int(seed) & (2 ** 64 - 1), int(first), int(n), pu, pp, pn))

    def _pointwise(self, fn, seed, first, n, pos_ratio, uid, iid, label):
        pu, nu, du, _ = _ids_arg(uid); pi, _, di, _ = _ids_arg(iid); pl, nl, dl, _ = _label_arg(label)
        assert du and di and dl and nu >= n and nl >= n, "the sampler writes device buffers"
        check(fn(self._h, int(seed) & (2 ** 64 - 1), int(first), int(n), float(pos_ratio), pu, pi, pl))

    def stratified_pointwise(self, seed, first, n, pos_ratio, uid, iid, label):
        """(user, item, label) samples [first, first + n) of `Dataset.stratified_pointwise` into DEVICE buffers; the stream
        is sequential (first = 0, then each call continues where the previous one stopped)."""
        self._pointwise(self._lib.orx_sampler_stratified, seed, first, n, pos_ratio, uid, iid, label)

    def per_pos_stratified_pointwise(self, seed, first, n, pos_ratio, uid, iid, label):
        """(user, item, label) samples [first, first + n) of `Dataset.per_pos_stratified_pointwise` into DEVICE buffers"""
        self._pointwise(self._lib.orx_sampler_per_pos_stratified, seed, first, n, pos_ratio, uid, iid, label)
