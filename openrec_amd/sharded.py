"""Row-sharded multi-GPU train step: one process per GPU,
`torch.distributed` collectives (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests), compute in libopenrec_hip.so.

Sharding: row r of every table lives on rank r % N at local index r // N.
One step over the GLOBAL batch (all ranks' triplets; the loss mean runs over
N*B) has exactly the single-GPU semantics:

  1. route   every triplet goes to the owner of its user row      all_to_all  (12 B / triplet)
  2. request the two item ids of each triplet go to their owners   all_to_all  (4 B / id)
  3. rows    owners gather row + bias and send them back           all_to_all  ((D+4)*4 B / id)
  4. local   gather user rows, score, loss, per-occurrence grads   (HIP: orx_gather_rows, orx_pair_grads)
  5. users   apply user-row gradients on the local shard           (HIP: orx_apply_rows)
  6. items   item-row gradients travel back along route 2          all_to_all  ((D+4)*4 B / id)
             and are applied by the owners                         (HIP: orx_apply_rows)

Reads (3) and writes (5, 6) are separate phases, so every gradient is taken on
the pre-step tables and duplicates need no special care beyond the optimizer's
own rule (SGD accumulates every occurrence, Adagrad sums duplicates first).
All exchanges use fixed-capacity buckets (no host synchronization, no size
exchange); `check()` reports a capacity overflow.
"""
from __future__ import annotations

import math

import numpy as np

import torch
import torch.distributed as dist


def owner_of(ids, world):
    return ids % world


def local_index(ids, world):
    return torch.div(ids, world, rounding_mode="floor")


def rows_on_rank(n_rows, rank, world):
    return (n_rows - rank + world - 1) // world


def bucket_slots(dest, world, cap):
    """dest: int64 [n], owner rank of each element or -1 (dead element).
    Returns (slot [n] int64 in [0, world*cap) or -1, overflow flag tensor).
    Elements keep their relative order inside a bucket (stable)."""
    n = dest.numel()
    dev = dest.device
    d = torch.where(dest >= 0, dest, torch.full_like(dest, world))
    order = torch.argsort(d, stable=True)
    ds = d[order]
    counts = torch.bincount(ds, minlength=world + 1)
    starts = torch.cumsum(counts, 0) - counts
    pos = torch.arange(n, device=dev) - starts[ds]
    ok = (ds < world) & (pos < cap)
    slot_sorted = torch.where(ok, ds * cap + pos, torch.full_like(ds, -1))
    slot = torch.empty_like(slot_sorted)
    slot[order] = slot_sorted
    overflow = (counts[:world] > cap).any()
    return slot, overflow


def make_comm(be, rank, world, group=None, rccl=None, vgroup=None):
    """-> an orx_comm handle on the context of backend `be` (anything with .ctx, .lib, ._ffi, .device).  `rccl`: exchange through
    RCCL (default: world > 1); the 128-byte id made by rank 0 reaches the other ranks through `group` (any torch.distributed
    backend).  `vgroup`: an orx_vgroup handle -- ranks in threads of this process on one device (tests)."""
    import ctypes
    import weakref
    if vgroup is not None:
        h = ctypes.c_void_p()
        be._ffi.check(be.lib.orx_comm_create_virtual(be.ctx._h, vgroup, rank, ctypes.byref(h)))
        weakref.finalize(be, be.lib.orx_comm_destroy, h)
        return h
    rccl = world > 1 if rccl is None else rccl
    idp = None
    if rccl:
        buf = torch.zeros(be._ffi.ORX_COMM_ID_BYTES, dtype=torch.uint8)
        err = None
        if rank == 0:
            try:
                raw = (ctypes.c_char * be._ffi.ORX_COMM_ID_BYTES)()
                be._ffi.check(be.lib.orx_comm_unique_id(ctypes.cast(raw, ctypes.c_void_p)))
                buf = torch.frombuffer(bytearray(raw.raw), dtype=torch.uint8).clone()
            except Exception as e:                    # noqa: BLE001  (the others wait in the broadcast: they get zeros)
                err = e
        if world > 1:
            dev_buf = buf.to(be.device) if dist.get_backend(group) == "nccl" else buf
            dist.broadcast(dev_buf, src=0, group=group)
            buf = dev_buf.cpu()
        if not bool(buf.any()):
            raise RuntimeError(f"rank 0 could not make an RCCL id: {err!r}")
        be._id_keep = buf.contiguous()
        idp = ctypes.c_void_p(be._id_keep.data_ptr())
    h = ctypes.c_void_p()
    be._ffi.check(be.lib.orx_comm_create(be.ctx._h, idp, rank, world, ctypes.byref(h)))
    weakref.finalize(be, be.lib.orx_comm_destroy, h)
    return h


class HipBackend:
    """Compute backend = the C ABI.  Buffers are torch tensors on the rank's GPU,
    kernels run on torch's current stream so that they order with the collectives."""

    def __init__(self, device, opt_kind, lr, opt_kw=None):
        from . import runtime as rt, _ffi
        self.rt, self._ffi = rt, _ffi
        self.device = device
        # A dedicated torch stream: the library enqueues its kernels on it and the step runs its
        # torch ops / collectives under it, so kernels and RCCL exchanges are ordered by the stream.
        # (torch's default stream has handle 0, which the C ABI reads as "create a private stream".)
        self.stream = torch.cuda.Stream(device=device)
        assert self.stream.cuda_stream != 0
        self.ctx = rt.Context(device.index if device.index is not None else 0, stream=self.stream.cuda_stream)
        kw = opt_kw or {}
        if opt_kind == "sgd":
            self.opt = rt.Optimizer.sgd(lr, ctx=self.ctx)
        elif opt_kind == "adagrad":
            self.opt = rt.Optimizer.adagrad(lr, kw.get("initial_accumulator_value", 0.1), kw.get("epsilon", 1e-7), ctx=self.ctx)
        elif opt_kind == "adam":
            # TF-2.0 dense-decay Adam, applied lazily by the library (rows replay their gradient-free steps when
            # next gathered or given a gradient); the step counter advances once per step, see begin_step()
            self.opt = rt.Optimizer.adam(lr, kw.get("beta_1", 0.9), kw.get("beta_2", 0.999), kw.get("epsilon", 1e-7), ctx=self.ctx)
        else:
            raise ValueError("sharded tables support sgd, adagrad and adam")
        self.opt_kind = opt_kind
        self.lib = self.ctx._lib

    def begin_step(self):
        """after the step's gathers, before its applies: Keras `iterations` += 1 (Adam's lr_t and the lazy replay)"""
        if self.opt_kind == "adam":
            self.opt.advance()           # (the engine's optimizer holds only the engine's tables)

    # checkpoints of this rank's shards: runtime.save_checkpoint's directory format, one set of files per rank
    def save_tables(self, path, tables, shard):
        self.stream.synchronize()
        self.rt.save_checkpoint(path, tables, self.opt, shard=shard)

    def load_tables(self, path, tables, shard):
        self.stream.synchronize()
        self.rt.load_checkpoint(path, tables, self.opt, shard=shard)

    def make_table(self, rows, dim, seed):
        return self.rt.Table(max(rows, 1), dim, self.ctx).init_uniform(seed=seed)

    def gather_rows(self, table, bias, ids, out):
        self._ffi.check(self.lib.orx_gather_rows(self.ctx._h, table._h, bias._h if bias is not None else None,
                                                 ids.data_ptr(), ids.numel(), out.data_ptr(), out.shape[1]))

    def pair_grads(self, model, dim, u, p, n, valid, b_global, margin, gu, gp, gn, accum):
        mid = {"bpr": self._ffi.ORX_BPR, "ucml": self._ffi.ORX_UCML}[model]
        self._ffi.check(self.lib.orx_pair_grads(self.ctx._h, mid, dim, u.data_ptr(), p.data_ptr(), n.data_ptr(),
                                                u.shape[1], valid.data_ptr(), valid.numel(), b_global, margin, 0,
                                                gu.data_ptr(), gp.data_ptr(), gn.data_ptr(), gu.shape[1],
                                                accum.data_ptr()))

    def apply_rows(self, table, bias, ids, grads):
        self._ffi.check(self.lib.orx_apply_rows(self.ctx._h, self.opt._h, table._h,
                                                bias._h if bias is not None else None,
                                                ids.data_ptr(), ids.numel(), grads.data_ptr(), grads.shape[1]))

    def rows_dupflags(self, table, ids2d, out):
        """ids2d [K, n] int32 (local row ids, < 0 = padding) -> out [K, n] uint8 duplicate flags per list"""
        K, n = ids2d.shape
        self._ffi.check(self.lib.orx_rows_dupflags(self.ctx._h, table.rows, ids2d.data_ptr(), K, n, n, out.data_ptr()))

    def apply_rows_flagged(self, table, bias, ids, grads, dflag):
        self._ffi.check(self.lib.orx_apply_rows_flagged(self.ctx._h, self.opt._h, table._h,
                                                        bias._h if bias is not None else None, ids.data_ptr(), ids.numel(),
                                                        grads.data_ptr(), grads.shape[1], dflag.data_ptr()))

    # ---- device-side exchange plan (fast path) ----------------------------
    fast = True

    def shard_route(self, uid, pid, nid, n_users, n_items, world, cap, send, counters, overflow):
        self._ffi.check(self.lib.orx_shard_route(self.ctx._h, uid.data_ptr(), pid.data_ptr(), nid.data_ptr(), uid.numel(),
                                                 n_users, n_items, world, cap, send.data_ptr(), counters.data_ptr(),
                                                 overflow.data_ptr()))

    def shard_request(self, trip, world, cap, send_ids, slot, u_loc, counters, overflow):
        self._ffi.check(self.lib.orx_shard_request(self.ctx._h, trip.data_ptr(), trip.shape[0], world, cap,
                                                   send_ids.data_ptr(), slot.data_ptr(), u_loc.data_ptr(),
                                                   counters.data_ptr(), overflow.data_ptr()))

    def shard_route_steps(self, uid, pid, nid, n_users, n_items, world, cap, send, counters, overflow):
        K, B = uid.shape
        self._ffi.check(self.lib.orx_shard_route_steps(self.ctx._h, uid.data_ptr(), pid.data_ptr(), nid.data_ptr(), K, B, uid.stride(0),
                                                       n_users, n_items, world, cap, send.data_ptr(), counters.data_ptr(),
                                                       overflow.data_ptr()))

    def shard_request_steps(self, trip, world, cap, send_ids, slot, u_loc, counters, overflow):
        K, T = trip.shape[0], trip.shape[1]
        self._ffi.check(self.lib.orx_shard_request_steps(self.ctx._h, trip.data_ptr(), K, T, world, cap, send_ids.data_ptr(),
                                                         slot.data_ptr(), u_loc.data_ptr(), counters.data_ptr(), overflow.data_ptr()))

    def shard_request_dedup_steps(self, trip, world, cap, n_items, send_ids, slot, u_loc, dupref, overflow):
        """the request plan with per-destination dedup: references of a list that ask for the same item share ONE slot
        (dupref = 1 on them); the distinct items of an owner fill its bucket in ascending row order.  -> what the gradient
        kernel needs to sum the shared slots (opaque: the sorted reference list and the list of shared slots)"""
        K, T = trip.shape[0], trip.shape[1]
        dev = trip.device
        plan = dict(sorted=torch.empty((K, 2 * T, 2), dtype=torch.int32, device=dev), seglist=torch.empty((K, T, 2), dtype=torch.int32, device=dev),
                    segcount=torch.empty(K, dtype=torch.int32, device=dev))
        self._ffi.check(self.lib.orx_shard_request_dedup_steps(self.ctx._h, trip.data_ptr(), K, T, world, cap, n_items, send_ids.data_ptr(),
                                                               slot.data_ptr(), u_loc.data_ptr(), dupref.data_ptr(), plan["sorted"].data_ptr(),
                                                               plan["seglist"].data_ptr(), plan["segcount"].data_ptr(), overflow.data_ptr()))
        return plan

    def _dedup_args(self, dupref, rows_in):
        """(dupref, sorted, seglist, segcount, gdup) pointers of one list, or five NULLs"""
        if dupref is None:
            return (None,) * 5
        flags, plan, l = dupref
        n2 = flags.shape[1]
        key = (n2, rows_in.shape[1])
        if getattr(self, "_gdup_key", None) != key:
            self._gdup = [torch.empty((n2, rows_in.shape[1]), dtype=torch.float32, device=rows_in.device) for _ in range(2)]
            self._gdup_key = key
        g = self._gdup[l % 2]                                # (the two halves of an overlapped step use one each)
        return (flags[l].data_ptr(), plan["sorted"][l].data_ptr(), plan["seglist"][l].data_ptr(), plan["segcount"][l:].data_ptr(), g.data_ptr())

    def shard_localize(self, ids, world, out):
        self._ffi.check(self.lib.orx_shard_localize(self.ctx._h, ids.data_ptr(), ids.numel(), world, out.data_ptr()))

    def shard_grads(self, model, user, rows_in, u_loc, slot, b_global, margin, gu, send_g, accum, dupref=None):
        """`dupref` = (flags [L, 2T], plan of shard_request_dedup_steps, list index): shared slots receive the sum of their references"""
        mid = {"bpr": self._ffi.ORX_BPR, "ucml": self._ffi.ORX_UCML}[model]
        self._ffi.check(self.lib.orx_shard_grads(self.ctx._h, mid, user._h, rows_in.data_ptr(), u_loc.data_ptr(),
                                                 slot.data_ptr(), *self._dedup_args(dupref, rows_in),
                                                 u_loc.numel(), rows_in.shape[1], b_global, margin, 0,
                                                 gu.data_ptr(), send_g.data_ptr(), accum.data_ptr()))

    def shard_grads_sgd(self, model, user, rows_in, u_loc, slot, dup_u, b_global, margin, gu, u_apply, send_g, accum, dupref=None):
        """gradients + SGD apply of the user rows referenced once (the duplicated ones are left in gu / u_apply)"""
        mid = {"bpr": self._ffi.ORX_BPR, "ucml": self._ffi.ORX_UCML}[model]
        self._ffi.check(self.lib.orx_shard_grads_sgd(self.ctx._h, mid, self.opt._h, user._h, rows_in.data_ptr(), u_loc.data_ptr(),
                                                     slot.data_ptr(), *self._dedup_args(dupref, rows_in),
                                                     dup_u.data_ptr(), u_loc.numel(), rows_in.shape[1], b_global, margin, 0,
                                                     gu.data_ptr(), u_apply.data_ptr(), send_g.data_ptr(), accum.data_ptr()))

    # ---- the whole K-step loop in the library (orx_sharded_pairwise_steps) --------
    def make_comm(self, rank, world, group=None, rccl=None, vgroup=None):
        return make_comm(self, rank, world, group, rccl, vgroup)

    def sharded_steps(self, comm, model, U, V, b, uid, pid, nid, n_users, n_items, margin, slack, plan_chunk, overlap, accum, ovf, dedup=True,
                      hot=None):
        """hot = (hot_items, Vh, bh): the K steps with the replicated hot items (orx_sharded_pairwise_steps_hot)"""
        mid = {"bpr": self._ffi.ORX_BPR, "ucml": self._ffi.ORX_UCML}[model]
        K, B = uid.shape
        assert uid.stride(1) == 1 and pid.stride() == uid.stride() and nid.stride() == uid.stride()
        flags = (self._ffi.ORX_SHARD_OVERLAP if overlap else 0) | \
            (0 if dedup is None else (self._ffi.ORX_SHARD_DEDUP if dedup else self._ffi.ORX_SHARD_NO_DEDUP))
        if hot is not None and hot[0] > 0:
            H, Vh, bh, cold = hot
            self._ffi.check(self.lib.orx_sharded_pairwise_steps_hot(comm, self.opt._h, mid, U._h, V._h, b._h, Vh._h, bh._h, int(H), float(cold),
                                                                    uid.data_ptr(), pid.data_ptr(), nid.data_ptr(), K, B, uid.stride(0),
                                                                    n_users, n_items, margin, slack, plan_chunk, flags, accum.data_ptr(),
                                                                    ovf.data_ptr()))
            return
        self._ffi.check(self.lib.orx_sharded_pairwise_steps(comm, self.opt._h, mid, U._h, V._h, b._h, uid.data_ptr(), pid.data_ptr(),
                                                            nid.data_ptr(), K, B, uid.stride(0), n_users, n_items, margin, slack,
                                                            plan_chunk, flags, accum.data_ptr(), ovf.data_ptr()))

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def check(self):
        self.ctx.check_index_error()

    def prof(self):
        return self.ctx.prof_get()


class ShardedPairwise:
    def __init__(self, model, opt, n_users, n_items, dim, lr, rank, world, device, seed=0, margin=0.5,
                 backend=None, slack=1.05, opt_kw=None, group=None, a2a_fn=None, fast=None, engine=None, dedup=None, vgroup=None,
                 hot_items=0, allreduce_fn=None, hot_cold_fraction=1.0):
        assert model in ("bpr", "ucml")
        self.model, self.dim, self.margin = model, dim, margin
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.n_users, self.n_items = n_users, n_items
        self.be = backend if backend is not None else HipBackend(device, opt, lr, opt_kw)
        self.opt_kind = opt
        self.U = self.be.make_table(rows_on_rank(n_users, rank, world), dim, seed * 3 + 0 + 1000 * rank)
        self.V = self.be.make_table(rows_on_rank(n_items, rank, world), dim, seed * 3 + 1 + 1000 * rank)
        self.b = self.be.make_table(rows_on_rank(n_items, rank, world), 1, seed * 3 + 2 + 1000 * rank)
        self.slack = slack
        self.DS = dim + 4                       # row + bias column, rows stay 16-B aligned
        self.accum = torch.zeros(2, dtype=torch.float64, device=device)
        self.overflow = torch.zeros((), dtype=torch.bool, device=device)
        self._cap_for = {}
        self.a2a_fn = a2a_fn
        self.force_collectives = False                    # world 1: still go through torch.distributed (smoke test)
        # the device-side plan needs the HIP building blocks and a float4 row path
        can_fast = getattr(self.be, "fast", False) and dim in (16, 32, 64, 128, 256)
        self.fast = can_fast if fast is None else (fast and can_fast)
        self._ovf = torch.zeros(1, dtype=torch.int32, device=device) if self.fast else None
        self._bufs = {}
        self.engine = engine                              # "python": K-step calls stay on the per-phase path
        # per-destination dedup of the item requests of the K-step plans (an item several references of a list ask for travels
        # once); None: decided per call from the list length (`_dedup_for`)
        self.dedup = dedup
        self._comm = None
        self.vgroup = vgroup                              # orx_vgroup handle: ranks in threads of one process (tests)
        # ---- skewed item popularity: the `hot_items` most popular items -- ids 0 .. hot_items-1, a vocabulary
        # sorted by frequency -- are REPLICATED on every rank (rows + biases, [H, D] + [H, 1]).  Their references read the local
        # replica and send nothing; their gradients are summed locally, then over the ranks by ONE all-reduce of the [H, D + 4]
        # block per step, and every rank applies the same sums to its replica (the replicas stay bit-identical: same addends,
        # same order).  With Zipf(1.05) ids over 1 M items the first 16 k items take 78 % of the item references of a step: that
        # share of the row and gradient exchanges leaves the wire, for 2 * (N - 1) / N * H * (D + 4) * 4 bytes of all-reduce.
        # The replica is the authority for those rows while training; sync_hot() writes them back into the owners' shards.
        # Per-phase path only (the K-step engines fall back to it); exact: TF sums the gradients of duplicate ids before the
        # sparse apply and a sum over ranks of per-rank sums is such a sum.
        self.hot = int(min(max(hot_items, 0), n_items))
        # the library's engine sizes the exchanged buckets for this share of a list's item references (those that are NOT hot): the
        # buckets travel whole, so this is what takes the hot rows off the wire; more cold references than that -> check() raises
        self.hot_cold_fraction = float(hot_cold_fraction)
        self.allreduce_fn = allreduce_fn
        if self.hot:
            # (the per-phase device plans do not know about replicas; the library's K-step engine does: orx_sharded_pairwise_steps_hot)
            self._fast_hot = bool(self.fast) and hasattr(self.be, "sharded_steps") and world < 64
            self.fast = False
            if not self._fast_hot:
                self._ovf = None
            self.Vh = self.be.make_table(self.hot, dim, seed * 3 + 7)
            self.bh = self.be.make_table(self.hot, 1, seed * 3 + 8)
            self._hot_loaded = False

    # ---- hot rows ------------------------------------------------------------------------------------------------------
    def _allreduce(self, x):
        if self.allreduce_fn is not None:                 # tests: in-place sum over an in-process fake cluster
            self.allreduce_fn(x)
        elif self.world > 1 or self.force_collectives:
            dist.all_reduce(x, group=self.group)
        return x

    def load_hot(self):
        """replicas <- the owners' shards (collective): row i < hot_items lives at local row i // world of rank i % world.
        Called once before the first step (and after anything wrote the shards: load(), a test's U.write ...)."""
        H, N, D, dev = self.hot, self.world, self.dim, self.device
        if not H:
            return
        blk = torch.zeros((H, D + 1), dtype=torch.float32, device=dev)
        mine = torch.arange(self.rank, H, N, device=dev)                  # the hot rows this rank owns
        if mine.numel():
            rows = torch.zeros((mine.numel(), self.DS), dtype=torch.float32, device=dev)
            self.be.gather_rows(self.V, self.b, local_index(mine, N).to(torch.int32).contiguous(), rows)
            if hasattr(self.be, "stream"):
                self.be.stream.synchronize()
            blk[mine] = rows[:, :D + 1]
        self._allreduce(blk)                                              # (every row has exactly one owner: the sum is a gather)
        self._write_table(self.Vh, blk[:, :D]); self._write_table(self.bh, blk[:, D:D + 1])
        self._hot_loaded = True

    def sync_hot(self):
        """the owners' shards <- replicas (local: every rank's replica is the same): after this, V / b hold the trained hot rows"""
        H, N, D = self.hot, self.world, self.dim
        if not H or not self._hot_loaded:
            return
        if hasattr(self.be, "stream"):
            self.be.stream.synchronize()
        vh, bh = self._read_table(self.Vh), self._read_table(self.bh)
        own = np.arange(self.rank, H, N)
        if own.size:
            v, b = self._read_table(self.V), self._read_table(self.b)
            v[own // N] = vh[own]; b[own // N] = bh[own]
            self._write_table(self.V, v); self._write_table(self.b, b)

    @staticmethod
    def _read_table(t):
        return t.read() if hasattr(t, "read") else t.w.copy()

    @staticmethod
    def _write_table(t, values):
        a = values.detach().cpu().numpy() if hasattr(values, "detach") else np.asarray(values)
        if hasattr(t, "write"):
            t.write(np.ascontiguousarray(a, np.float32))
        else:
            t.w[:] = a

    def _library_engine(self):
        """The C engine (orx_sharded_pairwise_steps) takes the K-step calls when the compute backend is the library and the
        exchange is RCCL's (process group backend "nccl") or the identity (one rank).  gloo groups and injected exchanges
        (tests) keep the per-phase path below, which drives the same kernels from here."""
        import os
        if self.engine is None and os.environ.get("ORX_SHARD_ENGINE") == "python":
            self.engine = "python"
        if self.engine == "python" or self.a2a_fn is not None or not hasattr(self.be, "sharded_steps") or (self.hot and not getattr(self, "_fast_hot", False)):
            return False
        if self._comm is None and self.vgroup is not None:
            self._comm = self.be.make_comm(self.rank, self.world, vgroup=self.vgroup)
        if self._comm is None:
            rccl = self.world > 1 or self.force_collectives
            if self.world > 1 and dist.get_backend(self.group) != "nccl":
                self.engine = "python"
                return False
            # every rank must end up on the same path: a rank that cannot make its communicator (librccl.so not loadable, ...)
            # takes all of them to the per-phase path over torch.distributed
            comm, err = None, None
            try:
                comm = self.be.make_comm(self.rank, self.world, self.group, rccl=rccl)
            except Exception as e:                        # noqa: BLE001
                err = e
            ok = torch.tensor([0 if comm is None else 1], dtype=torch.int32, device=self.device)
            if self.world > 1:
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
            if int(ok.item()) == 0:
                import warnings
                warnings.warn(f"sharded engine: no library communicator on every rank ({err!r}); using the per-phase path", RuntimeWarning)
                self.engine = "python"
                return False
            self._comm = comm
        return True

    # capacity of one (source, destination) bucket for n elements spread over `world` ranks
    def _cap(self, n):
        if n not in self._cap_for:
            mean = n / self.world
            self._cap_for[n] = int(math.ceil(mean * self.slack + 6 * math.sqrt(mean) + 16))
        return self._cap_for[n]

    def _a2a(self, send, recv=None):
        if self.a2a_fn is not None:                       # injected exchange (single-process tests)
            recv = torch.empty_like(send) if recv is None else recv
            self.a2a_fn(recv, send)
            return recv
        if self.world == 1 and not self.force_collectives:
            return send                                   # a one-rank exchange is the identity
        recv = torch.empty_like(send) if recv is None else recv
        dist.all_to_all_single(recv, send, group=self.group)
        return recv

    def _a2a_async(self, send, recv):
        """start an all-to-all; returns (recv, wait) -- wait() orders the caller's stream behind it.  RCCL runs it on the
        process group's own stream, so kernels enqueued before wait() overlap with the exchange."""
        if self.a2a_fn is not None or (self.world == 1 and not self.force_collectives):
            out = self._a2a(send, recv)
            if out is not recv:            # (a one-rank exchange is the identity: the overlapped path reads `recv`, halves side by side)
                recv.copy_(out)
            return recv, (lambda: None)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=True)
        return recv, work.wait

    # ------------------------------------------------------------------ fast path
    def _buffers(self, B, idx=0):
        """exchange buffers for steps of B triplets per rank (`idx`: a second, independent set for the overlapped halves)"""
        if (B, idx) not in self._bufs:
            N, dev, DS, D = self.world, self.device, self.DS, self.dim
            cap1 = self._cap(B); T = N * cap1
            cap2 = self._cap(2 * T); M = N * cap2
            i32 = dict(dtype=torch.int32, device=dev); f32 = dict(dtype=torch.float32, device=dev)
            self._bufs[(B, idx)] = dict(
                cap1=cap1, T=T, cap2=cap2, M=M,
                send1=torch.empty((T, 3), **i32), recv1=torch.empty((T, 3), **i32), cnt=torch.zeros(N, **i32),
                send2=torch.empty(M, **i32), req=torch.empty(M, **i32), req_loc=torch.empty(M, **i32),
                slot=torch.empty(2 * T, **i32), u_loc=torch.empty(T, **i32), u_apply=torch.empty(T, **i32),
                rows_out=torch.zeros((M, DS), **f32), rows_in=torch.empty((M, DS), **f32),
                gu=torch.zeros((T, D), **f32), send_g=torch.zeros((M, DS), **f32), g_in=torch.empty((M, DS), **f32))
        return self._bufs[(B, idx)]

    def _step_fast(self, uid, pid, nid):
        be, N = self.be, self.world
        B = uid.numel()
        f = self._buffers(B)
        be.shard_route(uid, pid, nid, self.n_users, self.n_items, N, f["cap1"], f["send1"], f["cnt"], self._ovf)
        mine = self._a2a(f["send1"], f["recv1"])                                   # 1. triplets -> user owner
        be.shard_request(mine, N, f["cap2"], f["send2"], f["slot"], f["u_loc"], f["cnt"], self._ovf)
        req = self._a2a(f["send2"], f["req"])                                      # 2. item ids -> item owner
        be.shard_localize(req, N, f["req_loc"])
        be.gather_rows(self.V, self.b, f["req_loc"], f["rows_out"])
        rows_in = self._a2a(f["rows_out"], f["rows_in"])                           # 3. item rows back
        be.shard_grads(self.model, self.U, rows_in, f["u_loc"], f["slot"], B * N, self.margin, f["gu"], f["send_g"],
                       self.accum)
        be.begin_step()
        be.apply_rows(self.U, None, f["u_loc"], f["gu"])                           # 5. user rows are local
        g_in = self._a2a(f["send_g"], f["g_in"])                                   # 6. item-row gradients -> owners
        be.apply_rows(self.V, self.b, f["req_loc"], g_in)
        return None

    # ---- checkpoints: every rank writes / reads its own shard (row r of a table = local row r // world of rank r % world) ----
    def save(self, path):
        """this rank's shards of the three tables (and their optimizer slots, the Adam step counter) into directory `path`
        (runtime.save_checkpoint: one .npy per tensor, streamed in row ranges; files carry `.rank<r>of<w>`).  Collective when
        world > 1: returns once every rank has written."""
        self.check()
        self.sync_hot()
        self.be.save_tables(path, dict(U=self.U, V=self.V, b=self.b), (self.rank, self.world))
        # The replicated hot rows are trained on the replica: their OPTIMIZER STATE (Adagrad accumulators, Adam m / v) lives in the
        # replica's slots, not in the owners' shards (sync_hot moves weights only).  Every rank writes its replica -- tables and slots,
        # identical on all ranks -- next to its shard, and a meta file says how the shard files are to be read.
        import json
        import os
        trained = bool(self.hot) and self._hot_loaded
        if trained:
            self.be.save_tables(os.path.join(path, "hot"), dict(Vh=self.Vh, bh=self.bh), (self.rank, self.world))
        with open(os.path.join(path, "sharded_meta.rank%dof%d.json" % (self.rank, self.world)), "w") as f:
            json.dump(dict(hot_items=self.hot, hot_saved=trained, opt=str(self.opt_kind), world=self.world), f)
        if self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)

    def load(self, path):
        """the inverse of save(): same world size and rank layout -- and, unless the optimizer is stateless, the same `hot_items`
        (the slots of the replicated rows come back with the replica, see save(); they cannot be re-sharded from here)"""
        import json
        import os
        meta_fn = os.path.join(path, "sharded_meta.rank%dof%d.json" % (self.rank, self.world))
        meta = dict(hot_items=0, hot_saved=False)               # (a checkpoint written before the meta file existed: no replicas)
        if os.path.exists(meta_fn):
            with open(meta_fn) as f:
                meta.update(json.load(f))
        saved_hot, hot_saved = int(meta["hot_items"]), bool(meta["hot_saved"])
        if not (saved_hot == self.hot or self.opt_kind == "sgd" or (self.hot == 0 and not hot_saved)):
            raise ValueError(f"checkpoint {path!r} was written with hot_items={saved_hot}, this engine has hot_items={self.hot}: the "
                             f"optimizer state of replicated rows ({self.opt_kind}) is stored with the replica and is not re-sharded")
        self.be.load_tables(path, dict(U=self.U, V=self.V, b=self.b), (self.rank, self.world))
        from_replica = bool(self.hot) and hot_saved and saved_hot == self.hot
        if from_replica:
            self.be.load_tables(os.path.join(path, "hot"), dict(Vh=self.Vh, bh=self.bh), (self.rank, self.world))
            self._hot_loaded = True
        if self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)
        if self.hot and not from_replica:
            self.load_hot()

    # ---- measurement of the library engine's exchanges (bench.py --gpus N) ----
    def comm_stats_start(self):
        """count the exchanges from here on (bytes on the wire, device time)"""
        if self._library_engine():
            self.be._ffi.check(self.be.lib.orx_comm_stats(self._comm, 1, None))

    def comm_stats_stop(self):
        """-> dict(exchanges, wire_bytes, exchange_ms, self_bytes) since comm_stats_start, or None without the library engine"""
        import ctypes
        if not self._library_engine():
            return None
        out = (ctypes.c_double * 4)()
        self.be._ffi.check(self.be.lib.orx_comm_stats(self._comm, 0, ctypes.cast(out, ctypes.c_void_p)))
        return dict(exchanges=int(out[0]), wire_bytes=float(out[1]), exchange_ms=float(out[2]), self_bytes=float(out[3]))

    def comm_ping(self, nbytes=32 << 20, reps=5):
        """all-to-all of `nbytes` per peer through the engine's own exchange (ncclSend / ncclRecv groups): what a link delivers
        -> dict(GBps_out, GBps_per_link, us_per_all_to_all); collective (every rank calls it)"""
        import ctypes
        if not self._library_engine():
            return None
        out = (ctypes.c_double * 3)()
        with self.be.stream_ctx():
            self.be._ffi.check(self.be.lib.orx_comm_ping(self._comm, int(nbytes), int(reps), ctypes.cast(out, ctypes.c_void_p)))
        return dict(GBps_out=float(out[0]), GBps_per_link=float(out[1]), us_per_all_to_all=float(out[2]))

    def steps(self, uid, pid, nid, plan_chunk=64, overlap=None):
        """K steps: uid/pid/nid int32 [K, B] on self.device.  The exchange PLAN of a step (which triplet goes
        to which user-owner, which item rows are requested from whom) depends on the ids alone, so it is made for
        `plan_chunk` steps at a time with ONE all-to-all per phase (routes 1 and 2 of the module docstring);
        the per-step dependency chain is then gather -> all-to-all (rows) -> gradients -> all-to-all (gradients)
        -> apply: two collectives per step instead of four."""
        K = uid.shape[0]
        if self.hot and getattr(self, "_fast_hot", False) and self._library_engine():
            # replicated hot items inside the library's K-step engine: the replicas are filled once, the engine keeps them identical
            if not self._hot_loaded:
                self.load_hot()
            ov = (self.world > 1 or self.force_collectives) if overlap is None else bool(overlap)
            with self.be.stream_ctx():
                self.be.sharded_steps(self._comm, self.model, self.U, self.V, self.b, uid, pid, nid, self.n_users, self.n_items,
                                      self.margin, self.slack, plan_chunk, ov, self.accum, self._ovf, dedup=False,
                                      hot=(self.hot, self.Vh, self.bh, self.hot_cold_fraction))
            return None
        if not self.fast:
            for k in range(K):
                self.step(uid[k], pid[k], nid[k])
            return None
        B = uid.shape[1]
        if self._library_engine():
            # one host call: plan, gathers, gradients, applies and the RCCL exchanges all run inside the library
            ov = (self.world > 1 or self.force_collectives) if overlap is None else bool(overlap)
            with self.be.stream_ctx():
                self.be.sharded_steps(self._comm, self.model, self.U, self.V, self.b, uid, pid, nid, self.n_users, self.n_items,
                                      self.margin, self.slack, plan_chunk, ov, self.accum, self._ovf,
                                      dedup=self.dedup)      # (None: the engine decides per list, by the same rule as _dedup_for)
            return None
        if overlap is None:       # two half-batches per step pay once there is a link to hide behind
            overlap = (self.world > 1 or self.force_collectives) and self.a2a_fn is None
        overlap = bool(overlap) and B % 2 == 0 and B >= 2
        with self.be.stream_ctx():
            for k0 in range(0, K, plan_chunk):
                sl = slice(k0, k0 + plan_chunk)
                if overlap:
                    self._steps_overlapped(uid[sl], pid[sl], nid[sl])
                else:
                    self._steps_planned(uid[sl], pid[sl], nid[sl])
        return None

    def _steps_overlapped(self, uid, pid, nid):
        """The planned K-step path with every step cut into two HALF-BATCHES whose exchanges overlap with the other half's
        kernels.  Within a step nothing depends on anything but the pre-step tables until the applies, so the order is
            gather A | a2a rows A || gather B | a2a rows B || grads A | a2a grads A || grads B | a2a grads B || apply V(A) | apply V(B)
        (|| = runs concurrently: the collectives are asynchronous on RCCL's stream, the kernels on the engine's).  All
        gathers and all gradient kernels of a step precede its applies (TF's snapshot semantics); the duplicate flags of the
        two apply lists are taken over BOTH halves together, so a row the halves share is still treated as duplicated.
        The next step's gathers need this step's applies: steps do not overlap with each other."""
        be, N = self.be, self.world
        Kc, B = uid.shape
        H = B // 2
        fA, fB = dict(self._buffers(H, 0)), dict(self._buffers(H, 1))
        T, M = fA["T"], fA["M"]
        if ("ov", H) not in self._bufs:     # the halves' gradient buffers side by side: Adagrad / Adam apply a step's list in ONE call
            f32 = dict(dtype=torch.float32, device=self.device)
            self._bufs[("ov", H)] = (torch.zeros((2 * T, self.dim), **f32), torch.empty((2 * M, self.DS), **f32))
        gu2, g_in2 = self._bufs[("ov", H)]
        fA["gu"], fB["gu"], fA["g_in"], fB["g_in"] = gu2[:T], gu2[T:], g_in2[:M], g_in2[M:]
        i32 = dict(dtype=torch.int32, device=self.device)
        # plan 2 Kc half-steps: half h of step k is list 2k + h
        uh, ph, nh = (x.contiguous().reshape(2 * Kc, H) for x in (uid, pid, nid))
        send1 = torch.empty((2 * Kc, T, 3), **i32); cnt = torch.empty((2 * Kc, N), **i32)
        be.shard_route_steps(uh, ph, nh, self.n_users, self.n_items, N, fA["cap1"], send1, cnt, self._ovf)
        mine = self._a2a_steps(send1, N).contiguous()
        send2 = torch.empty((2 * Kc, M), **i32); slot = torch.empty((2 * Kc, 2 * T), **i32); u_loc = torch.empty((2 * Kc, T), **i32)
        dupref = self._request_plan(mine, fA["cap2"], send2, slot, u_loc, cnt)
        dr = (lambda l: {"dupref": (dupref[0], dupref[1], l)}) if dupref is not None else (lambda l: {})
        req = self._a2a_steps(send2, N)
        req_loc = torch.empty_like(req)
        be.shard_localize(req.reshape(-1), N, req_loc.reshape(-1))
        flagged = self.opt_kind == "sgd" and hasattr(be, "rows_dupflags")
        if flagged:        # flags over the two halves of a step TOGETHER ([Kc, 2T] / [Kc, 2M] views of the same memory)
            fu = torch.empty((2 * Kc, T), dtype=torch.uint8, device=self.device)
            fv = torch.empty((2 * Kc, M), dtype=torch.uint8, device=self.device)
            be.rows_dupflags(self.U, u_loc.view(Kc, 2 * T), fu.view(Kc, 2 * T))
            be.rows_dupflags(self.V, req_loc.view(Kc, 2 * M), fv.view(Kc, 2 * M))
        folded = flagged and hasattr(be, "shard_grads_sgd")
        for k in range(Kc):
            a, b = 2 * k, 2 * k + 1
            be.gather_rows(self.V, self.b, req_loc[a], fA["rows_out"])
            rows_a, wait_ra = self._a2a_async(fA["rows_out"], fA["rows_in"])
            be.gather_rows(self.V, self.b, req_loc[b], fB["rows_out"])
            rows_b, wait_rb = self._a2a_async(fB["rows_out"], fB["rows_in"])
            waits = []
            for h, f, rows, wait_rows in ((a, fA, rows_a, wait_ra), (b, fB, rows_b, wait_rb)):
                wait_rows()
                if folded:
                    be.shard_grads_sgd(self.model, self.U, rows, u_loc[h], slot[h], fu[h], B * N, self.margin, f["gu"], f["u_apply"],
                                       f["send_g"], self.accum, **dr(h))
                else:
                    be.shard_grads(self.model, self.U, rows, u_loc[h], slot[h], B * N, self.margin, f["gu"], f["send_g"], self.accum, **dr(h))
                waits.append(self._a2a_async(f["send_g"], f["g_in"]))
            be.begin_step()
            if flagged:                                                      # SGD: every occurrence accumulates, half by half
                for h, f in ((a, fA), (b, fB)):                              # user rows are local
                    be.apply_rows_flagged(self.U, None, f["u_apply"] if folded else u_loc[h], f["gu"], fu[h])
                for h, (g_in, wait_g) in zip((a, b), waits):                 # item-row gradients at their owners
                    wait_g()
                    be.apply_rows_flagged(self.V, self.b, req_loc[h], g_in, fv[h])
            else:                                                            # Adagrad / Adam sum a row's duplicates FIRST: one list per step
                be.apply_rows(self.U, None, u_loc.view(Kc, 2 * T)[k], gu2)
                for _, wait_g in waits:
                    wait_g()
                be.apply_rows(self.V, self.b, req_loc.view(Kc, 2 * M)[k], g_in2)

    def _dedup_for(self, b_list):
        """Dedup costs a sort and an un-sort of the references at plan time: on where the item references a rank handles per list
        (2 x b_list: b_list triplets reach it) are at least half as many as the items -- most slots are then shared --, off for
        sparse lists (the library engine decides the same way)."""
        if not (self.fast and hasattr(self.be, "shard_request_dedup_steps")):
            return False
        if self.dedup is not None:
            return bool(self.dedup)
        return 4 * b_list >= self.n_items

    def _request_plan(self, mine, cap2, send2, slot, u_loc, cnt):
        """route 2 of the lists in `mine` [L, T, 3]; -> the shared-slot flags [L, 2T] (dedup) or None"""
        be, N = self.be, self.world
        if not self._dedup_for(mine.shape[1]):
            be.shard_request_steps(mine, N, cap2, send2, slot, u_loc, cnt, self._ovf)
            return None
        dupref = torch.empty(slot.shape, dtype=torch.uint8, device=self.device)
        plan = be.shard_request_dedup_steps(mine, N, cap2, self.n_items, send2, slot, u_loc, dupref, self._ovf)
        return dupref, plan

    def _a2a_steps(self, x, N):
        """x: [Kc, N * c, ...] per-step buckets -> the same layout after ONE all-to-all over all Kc steps"""
        if self.world == 1 and not self.force_collectives and self.a2a_fn is None:
            return x
        Kc = x.shape[0]
        c = x.shape[1] // N
        tail = x.shape[2:]
        send = x.reshape(Kc, N, c, *tail).transpose(0, 1).contiguous().reshape(N * Kc * c, *tail)     # [dest][step][slot]
        recv = self._a2a(send)
        return recv.reshape(N, Kc, c, *tail).transpose(0, 1).contiguous().reshape(Kc, N * c, *tail)   # [step][src][slot]

    def _steps_planned(self, uid, pid, nid):
        be, N = self.be, self.world
        Kc, B = uid.shape
        f = self._buffers(B)
        T, M = f["T"], f["M"]
        i32 = dict(dtype=torch.int32, device=self.device)
        send1 = torch.empty((Kc, T, 3), **i32)
        cnt = torch.empty((Kc, N), **i32)
        uid, pid, nid = uid.contiguous(), pid.contiguous(), nid.contiguous()
        be.shard_route_steps(uid, pid, nid, self.n_users, self.n_items, N, f["cap1"], send1, cnt, self._ovf)
        mine = self._a2a_steps(send1, N).contiguous()                              # 1. triplets -> user owner, all steps
        send2 = torch.empty((Kc, M), **i32); slot = torch.empty((Kc, 2 * T), **i32); u_loc = torch.empty((Kc, T), **i32)
        dupref = self._request_plan(mine, f["cap2"], send2, slot, u_loc, cnt)
        dr = (lambda l: {"dupref": (dupref[0], dupref[1], l)}) if dupref is not None else (lambda l: {})
        req = self._a2a_steps(send2, N)                                            # 2. item ids -> item owner, all steps
        req_loc = torch.empty_like(req)
        be.shard_localize(req.reshape(-1), N, req_loc.reshape(-1))
        # SGD: the duplicate flags of every step's two apply lists, one launch each for the whole chunk
        flagged = self.opt_kind == "sgd" and hasattr(be, "rows_dupflags")
        if flagged:
            fu = torch.empty((Kc, T), dtype=torch.uint8, device=self.device)
            fv = torch.empty((Kc, M), dtype=torch.uint8, device=self.device)
            be.rows_dupflags(self.U, u_loc, fu)
            be.rows_dupflags(self.V, req_loc, fv)
        for k in range(Kc):
            be.gather_rows(self.V, self.b, req_loc[k], f["rows_out"])
            rows_in = self._a2a(f["rows_out"], f["rows_in"])                       # 3. item rows back
            if flagged and hasattr(be, "shard_grads_sgd"):
                # 4 + 5: user rows referenced once are updated by the gradient kernel itself; the duplicated ones follow
                be.shard_grads_sgd(self.model, self.U, rows_in, u_loc[k], slot[k], fu[k], B * N, self.margin, f["gu"], f["u_apply"],
                                   f["send_g"], self.accum, **dr(k))
                be.begin_step()
                be.apply_rows_flagged(self.U, None, f["u_apply"], f["gu"], fu[k])
            else:
                be.shard_grads(self.model, self.U, rows_in, u_loc[k], slot[k], B * N, self.margin, f["gu"], f["send_g"], self.accum, **dr(k))
                be.begin_step()
                if flagged:
                    be.apply_rows_flagged(self.U, None, u_loc[k], f["gu"], fu[k])      # 5. user rows are local
                else:
                    be.apply_rows(self.U, None, u_loc[k], f["gu"])
            g_in = self._a2a(f["send_g"], f["g_in"])                               # 6. item-row gradients -> owners
            if flagged:
                be.apply_rows_flagged(self.V, self.b, req_loc[k], g_in, fv[k])
            else:
                be.apply_rows(self.V, self.b, req_loc[k], g_in)

    def step(self, uid, pid, nid):
        """uid/pid/nid: int32 [B] on self.device -- this rank's slice of the global batch."""
        if hasattr(self.be, "stream_ctx"):
            with self.be.stream_ctx():
                return self._step(uid, pid, nid)
        return self._step(uid, pid, nid)

    def _step(self, uid, pid, nid):
        if self.fast:
            return self._step_fast(uid, pid, nid)
        N, dev, DS, D = self.world, self.device, self.DS, self.dim
        B = uid.numel()
        b_global = B * N
        # ---- 1. route triplets to the owner of the user row
        cap1 = self._cap(B)
        trip = torch.stack([uid, pid, nid], 1).to(torch.int32)
        slot1, ov1 = bucket_slots(owner_of(uid.long(), N), N, cap1)
        send1 = torch.full((N * cap1 + 1, 3), -1, dtype=torch.int32, device=dev)
        send1.index_copy_(0, torch.where(slot1 >= 0, slot1, torch.full_like(slot1, N * cap1)), trip)
        mine = self._a2a(send1[:N * cap1].contiguous())                  # [T, 3]
        T = N * cap1
        u_g, p_g, n_g = mine[:, 0].long(), mine[:, 1].long(), mine[:, 2].long()
        live = u_g >= 0
        u_loc = torch.where(live, local_index(u_g, N), torch.full_like(u_g, -1)).to(torch.int32).contiguous()
        # ---- 2. request item rows from their owners (references to replicated hot items ask nobody)
        item_g = torch.cat([p_g, n_g])                                    # [2T]
        live2 = torch.cat([live, live])
        H = self.hot
        is_hot = live2 & (item_g < H) if H else torch.zeros_like(live2)
        if H and not self._hot_loaded:
            self.load_hot()
        cap2 = self._cap(2 * T)
        slot2, ov2 = bucket_slots(torch.where(live2 & ~is_hot, owner_of(item_g, N), torch.full_like(item_g, -1)), N, cap2)
        trash2 = N * cap2
        slot2s = torch.where(slot2 >= 0, slot2, torch.full_like(slot2, trash2))
        send2 = torch.full((trash2 + 1,), -1, dtype=torch.int32, device=dev)
        send2.index_copy_(0, slot2s, torch.where(is_hot, torch.full_like(item_g, -1), item_g).to(torch.int32))
        req = self._a2a(send2[:trash2].contiguous()).long()               # ids requested from me
        req_loc = torch.where(req >= 0, local_index(req, N), torch.full_like(req, -1)).to(torch.int32).contiguous()
        # ---- 3. owners gather rows (+ bias at column D) and send them back
        rows_out = torch.zeros((trash2, DS), dtype=torch.float32, device=dev)
        self.be.gather_rows(self.V, self.b, req_loc, rows_out)
        rows_in = torch.zeros((trash2 + 1, DS), dtype=torch.float32, device=dev)
        rows_in[:trash2] = self._a2a(rows_out)
        item_rows = rows_in.index_select(0, slot2s)                       # [2T, DS] in my reference order
        if H:                                                             # hot references: the local replica
            hot_ids = torch.where(is_hot, item_g, torch.full_like(item_g, -1)).to(torch.int32).contiguous()
            hot_rows = torch.zeros((2 * T, DS), dtype=torch.float32, device=dev)
            self.be.gather_rows(self.Vh, self.bh, hot_ids, hot_rows)
            item_rows = torch.where(is_hot[:, None], hot_rows, item_rows)
        p_rows, n_rows = item_rows[:T].contiguous(), item_rows[T:].contiguous()
        # a triplet whose item request overflowed a bucket is dropped (and reported by check())
        got = (slot2 >= 0) | is_hot
        ok = live & got[:T] & got[T:]
        valid = torch.where(ok, u_loc, torch.full_like(u_loc, -1)).contiguous()
        # ---- 4. local rows, score, gradients
        u_rows = torch.zeros((T, DS), dtype=torch.float32, device=dev)
        self.be.gather_rows(self.U, None, valid, u_rows)
        gu = torch.zeros((T, DS), dtype=torch.float32, device=dev)
        gp = torch.zeros((T, DS), dtype=torch.float32, device=dev)
        gn = torch.zeros((T, DS), dtype=torch.float32, device=dev)
        self.be.pair_grads(self.model, D, u_rows, p_rows, n_rows, valid, b_global, self.margin, gu, gp, gn, self.accum)
        # ---- 5. user rows are local
        self.be.begin_step()
        self.be.apply_rows(self.U, None, valid, gu)
        # ---- 6. item-row gradients go back along route 2 and are applied by the owners
        g_all = torch.cat([gp, gn])
        send_g = torch.zeros((trash2 + 1, DS), dtype=torch.float32, device=dev)
        dead = ~torch.cat([ok, ok]) | is_hot
        send_g.index_copy_(0, torch.where(dead, torch.full_like(slot2s, trash2), slot2s), g_all)
        g_in = self._a2a(send_g[:trash2].contiguous())
        # a request whose triplet was dropped carries a zero gradient: harmless for SGD, and for
        # Adagrad acc += 0, var -= 0
        self.be.apply_rows(self.V, self.b, req_loc, g_in)
        if H:
            # ---- 7. hot rows: this rank's gradients summed per row, ONE all-reduce of the [H, DS] block, the same apply on every replica
            # (every row of the block: a row nobody referenced carries a zero gradient, see above)
            hg = torch.zeros((H, DS), dtype=torch.float32, device=dev)
            use = is_hot & torch.cat([ok, ok])
            hg.index_add_(0, torch.where(use, item_g, torch.zeros_like(item_g)), torch.where(use[:, None], g_all, torch.zeros_like(g_all)))
            self._allreduce(hg)
            self.be.apply_rows(self.Vh, self.bh, torch.arange(H, dtype=torch.int32, device=dev), hg)
        self.overflow |= ov1 | ov2
        return None

    # ---- results -----------------------------------------------------------
    def loss_sums(self):
        """(sum over steps of loss, sum of l2_loss) over the global batch so far."""
        if hasattr(self.be, "stream"):
            self.be.stream.synchronize()
        t = self.accum.clone()
        if self.world > 1 and self.a2a_fn is None:          # (an injected exchange -- tests -- has no process group: per-rank sums)
            dist.all_reduce(t, group=self.group)
        return float(t[0]), float(t[1])

    def check(self):
        """Raise if an id was out of range or an exchange bucket overflowed (synchronizes)."""
        if hasattr(self.be, "stream"):
            self.be.stream.synchronize()
        if hasattr(self.be, "check"):
            self.be.check()
        ov = self.overflow.clone().to(torch.int32)
        if self._ovf is not None:
            ov = ov + self._ovf[0]
        if self.world > 1 and self.a2a_fn is None:
            dist.all_reduce(ov, group=self.group)
        if int(ov) != 0:
            raise RuntimeError("sharded exchange: bucket capacity exceeded (raise `slack`)")

    def prof(self):
        return self.be.prof() if hasattr(self.be, "prof") else {}
