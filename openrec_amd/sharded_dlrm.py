"""Hybrid-parallel DLRM train step: one process per GPU.

* the 26 embedding tables = ONE combined table, row-sharded: combined row r lives
  on rank r % N at local index r // N (model parallel);
* bottom / top MLPs and the feature interaction are replicated, every rank runs
  its own B samples (data parallel); the loss mean runs over the global batch N*B.

One step (exactly the single-GPU semantics of orx_dlrm_step on the concatenated
global batch; reference: recommenders/dlrm.py:63-74 + tf2_examples/dlrm_criteo.py:42-48):

  1. request  combined row ids of the B*n_emb lookups -> their owners       all_to_all (4 B / id)
  2. rows     owners gather the rows and send them back                     all_to_all (4*d B / id)
  3. local    forward + backward with the received rows (orx_dlrm_grads)
  4. dense    ONE all-reduce(sum) of the packed MLP gradients, then the dense optimizer rule
              on every rank (identical replicas stay identical)
  5. sparse   d loss / d row travels back along route 1                     all_to_all (4*d B / id)
              and the owners apply it (orx_apply_rows: SGD accumulates every
              occurrence, Adagrad / Adam sum duplicates first - as on one GPU)

Exchanges use fixed-capacity buckets (no size exchange, no host sync);
`check()` reports a capacity overflow.  Compute is libopenrec_hip.so through
`HipDLRMBackend`; the tests inject their own CPU backend for the gloo runs.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

from .sharded import bucket_slots, make_comm, rows_on_rank


class HipDLRMBackend:
    """Compute backend = the C ABI on this rank's GPU.  Kernels run on a dedicated torch
    stream under which the step also issues its torch ops and collectives."""

    def __init__(self, device, cfg, opt_kind, lr, opt_kw=None, seed=0, fp16_mlp=False):
        from . import runtime as rt, _ffi
        self.rt, self._ffi, self.device = rt, _ffi, device
        self.stream = torch.cuda.Stream(device=device)
        assert self.stream.cuda_stream != 0         # handle 0 would mean "create a private stream" to the C ABI
        self.ctx = rt.Context(device.index if device.index is not None else 0, stream=self.stream.cuda_stream)
        kw = opt_kw or {}
        if opt_kind == "sgd":
            self.opt = rt.Optimizer.sgd(lr, ctx=self.ctx)
        elif opt_kind == "adagrad":
            self.opt = rt.Optimizer.adagrad(lr, kw.get("initial_accumulator_value", 0.1), kw.get("epsilon", 1e-7), ctx=self.ctx)
        elif opt_kind == "adam":
            self.opt = rt.Optimizer.adam(lr, kw.get("beta_1", 0.9), kw.get("beta_2", 0.999), kw.get("epsilon", 1e-7), ctx=self.ctx)
        else:
            raise ValueError("unknown optimizer %r" % opt_kind)
        self.model = rt.DLRMModel(ctx=self.ctx, no_emb=True, seed=seed, fp16_mlp=fp16_mlp, **cfg)
        self.lib = self.ctx._lib

    def make_table(self, rows, dim, seed):
        return self.rt.Table(max(rows, 1), dim, self.ctx).init_uniform(seed=seed)

    def write_table(self, table, values):
        table.write(np.ascontiguousarray(values, np.float32))

    def read_table(self, table):
        return table.read()

    def gather_rows(self, table, ids, out):
        self._ffi.check(self.lib.orx_gather_rows(self.ctx._h, table._h, None, ids.data_ptr(), ids.numel(),
                                                 out.data_ptr(), out.shape[1]))

    def apply_rows(self, table, ids, grads):
        self._ffi.check(self.lib.orx_apply_rows(self.ctx._h, self.opt._h, table._h, None, ids.data_ptr(), ids.numel(),
                                                grads.data_ptr(), grads.shape[1]))

    def bucket(self, ids, world, cap, send_ids, slot, counters, overflow):
        self._ffi.check(self.lib.orx_shard_bucket(self.ctx._h, ids.data_ptr(), ids.numel(), world, cap, send_ids.data_ptr(),
                                                  slot.data_ptr(), counters.data_ptr(), overflow.data_ptr()))

    def localize(self, ids, world, out):
        self._ffi.check(self.lib.orx_shard_localize(self.ctx._h, ids.data_ptr(), ids.numel(), world, out.data_ptr()))

    def grads(self, dense, emb_rows, label, global_b, emb_grads, loss_accum):
        self.model.grads(dense.data_ptr(), emb_rows.data_ptr(), label.data_ptr(), label.numel(), global_b,
                         emb_grads.data_ptr(), loss_accum.data_ptr())

    def direct_ok(self):
        """may the step read the exchanged rows in place (orx_dlrm_grads_indirect)?"""
        return bool(self.lib.orx_dlrm_direct_ok(self.model._h))

    def grads_indirect(self, dense, rows, idx, label, global_b, grads_dst, loss_accum):
        """forward + backward with the embedding rows read in place from `rows` (the receive buffer) through idx [B, n_emb + 1];
        the gradient of every lookup is written to row idx[b, f] of `grads_dst` (the buffer that travels back)"""
        self._ffi.check(self.lib.orx_dlrm_grads_indirect(self.model._h, dense.data_ptr(), rows.data_ptr(), rows.shape[0], idx.data_ptr(),
                                                         label.data_ptr(), label.numel(), global_b, grads_dst.data_ptr(), loss_accum.data_ptr()))

    def dense_count(self):
        return self.model.dense_count()

    def dense_pack(self, flat):
        self.model.dense_pack(flat.data_ptr())

    def dense_apply(self, flat):
        self.model.dense_apply(self.opt, flat.data_ptr())

    def dense_param(self, kind, layer):
        return self.model.param(kind, layer)

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def check(self):
        self.ctx.check_index_error()

    # the whole K-step loop in the library (orx_sharded_dlrm_steps)
    def make_comm(self, rank, world, group=None, rccl=None, vgroup=None):
        return make_comm(self, rank, world, group, rccl, vgroup)

    def sharded_steps(self, comm, emb, dense, sparse, label, slack, loss_accum, ovf):
        K, B = label.shape
        self._ffi.check(self.lib.orx_sharded_dlrm_steps(comm, self.model._h, self.opt._h, emb._h, dense.data_ptr(), sparse.data_ptr(), label.data_ptr(),
                                                        K, B, slack, loss_accum.data_ptr(), ovf.data_ptr()))


class ShardedDLRM:
    def __init__(self, m_spa, ln_emb, ln_bot, ln_top, dense_dim, rank, world, device, opt="sgd", lr=0.01, opt_kw=None,
                 seed=0, slack=1.25, backend=None, group=None, a2a_fn=None, allreduce_fn=None, fp16_mlp=False, engine=None, vgroup=None,
                 **model_kw):
        self.m_spa, self.ln_emb, self.n_emb = int(m_spa), [int(x) for x in ln_emb], len(ln_emb)
        self.dense_dim, self.rank, self.world, self.device = int(dense_dim), rank, world, device
        self.group, self.a2a_fn, self.allreduce_fn, self.slack = group, a2a_fn, allreduce_fn, slack
        cfg = dict(m_spa=m_spa, ln_emb=ln_emb, ln_bot=ln_bot, ln_top=ln_top, dense_dim=dense_dim, **model_kw)
        self.be = backend if backend is not None else HipDLRMBackend(device, cfg, opt, lr, opt_kw, seed, fp16_mlp)
        self.total_rows = int(sum(self.ln_emb))
        self.offsets = torch.tensor(np.concatenate([[0], np.cumsum(self.ln_emb)[:-1]]), dtype=torch.int64, device=device)
        self.rows_t = torch.tensor(self.ln_emb, dtype=torch.int64, device=device)
        self.local_rows = rows_on_rank(self.total_rows, rank, world)
        self.emb = self.be.make_table(self.local_rows, self.m_spa, seed + 1000 + rank)
        self.n_dense = self.be.dense_count()
        self.flat = torch.zeros(self.n_dense, dtype=torch.float32, device=device)
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=device)
        self.overflow = torch.zeros((), dtype=torch.bool, device=device)
        self.bad_id = torch.zeros((), dtype=torch.bool, device=device)
        self.force_collectives = False          # world size 1 still goes through RCCL (bench.py --sharded under torchrun)
        self.copying = False                    # True: always the copying form of the local step (orx_dlrm_grads)
        self._cnt = torch.zeros(64, dtype=torch.int32, device=device)
        self._ovf = torch.zeros(1, dtype=torch.int32, device=device)
        self.engine = engine                    # None: the library's K-step engine where it applies; "python": the per-phase path
        self.vgroup = vgroup                    # orx_vgroup handle: ranks in threads of one process (tests)
        self._comm = None

    # ---- table access by GLOBAL combined row (tests, checkpoints) ----------
    def load_embeddings(self, combined):
        """combined: [sum(ln_emb), m_spa] array; this rank keeps rows rank, rank + N, ..."""
        self.be.write_table(self.emb, np.asarray(combined)[self.rank::self.world])

    def local_embeddings(self):
        return self.be.read_table(self.emb)[:self.local_rows]

    # ---- collectives ---------------------------------------------------------
    def _a2a(self, x):
        if self.a2a_fn is not None:            # tests: fn(recv, send), e.g. an in-process fake cluster
            out = torch.empty_like(x)
            self.a2a_fn(out, x)
            return out
        if self.world == 1 and not self.force_collectives:
            return x
        out = torch.empty_like(x)
        dist.all_to_all_single(out, x, group=self.group)
        return out

    def _allreduce(self, x):
        if self.allreduce_fn is not None:      # tests: in-place sum over the fake cluster
            self.allreduce_fn(x)
        elif self.world > 1 or self.force_collectives:
            dist.all_reduce(x, group=self.group)
        return x

    def _library_engine(self):
        """orx_sharded_dlrm_steps takes the K-step calls when the compute backend is the library, the model's shapes allow reading
        the exchanged rows in place, and the exchange is RCCL's (process group backend "nccl"), a virtual group, or the identity
        (one rank).  gloo groups and injected exchanges (tests) keep the per-phase path, which drives the same kernels from here."""
        import os
        if self.engine is None and os.environ.get("ORX_SHARD_ENGINE") == "python":
            self.engine = "python"
        if self.engine == "python" or self.a2a_fn is not None or self.allreduce_fn is not None or not hasattr(self.be, "sharded_steps") \
                or self.copying or not self.be.direct_ok():
            return False
        if self._comm is None and self.vgroup is not None:
            self._comm = self.be.make_comm(self.rank, self.world, vgroup=self.vgroup)
        if self._comm is None:
            rccl = self.world > 1 or self.force_collectives
            if self.world > 1 and dist.get_backend(self.group) != "nccl":
                self.engine = "python"
                return False
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
                warnings.warn(f"sharded DLRM: no library communicator on every rank ({err!r}); using the per-phase path", RuntimeWarning)
                self.engine = "python"
                return False
            self._comm = comm
        return True

    def steps(self, dense, sparse, label):
        """K steps: dense [K, B, dense_dim] fp32, sparse [K, B, n_emb] int32, label [K, B] fp32 -- this rank's slices of K global
        batches, on self.device.  One host call into the library's engine where it applies (see _library_engine), K calls of
        step() otherwise."""
        K = label.shape[0]
        if self._library_engine():
            self.be.stream.wait_stream(torch.cuda.current_stream(self.device))      # the inputs were made on the caller's stream
            with self.be.stream_ctx():
                de = dense.to(torch.float32).contiguous(); sp = sparse.to(torch.int32).contiguous(); la = label.to(torch.float32).contiguous()
                self.be.sharded_steps(self._comm, self.emb, de, sp, la, float(self.slack), self.loss_accum, self._ovf)
            return None
        for k in range(K):
            self.step(dense[k], sparse[k], label[k])
        return None

    def step(self, dense, sparse, label):
        """dense [B, dense_dim] fp32, sparse [B, n_emb] int (id within its own table), label [B] fp32 -
        this rank's slice of the global batch, on self.device."""
        if hasattr(self.be, "stream_ctx"):
            with self.be.stream_ctx():
                return self._step(dense, sparse, label)
        return self._step(dense, sparse, label)

    def _step(self, dense, sparse, label):
        N, dev, d, nf = self.world, self.device, self.m_spa, self.n_emb
        B = label.numel()
        dense = dense.to(torch.float32).contiguous()
        label = label.to(torch.float32).contiguous()
        sp = sparse.to(torch.int64)
        ok = (sp >= 0) & (sp < self.rows_t)                                   # the reference's gather raises on these
        self.bad_id |= (~ok).any()
        g = torch.where(ok, sp + self.offsets, torch.full_like(sp, -1)).reshape(-1)          # combined row ids [B*nf]
        # ---- 1. requests to the owners
        n = g.numel()
        cap = int(math.ceil(n / N * self.slack)) + 8
        trash = N * cap
        if hasattr(self.be, "bucket"):                                     # device-side plan (ballot + popcount slot claims)
            g32 = g.to(torch.int32).contiguous()
            send = torch.empty((trash,), dtype=torch.int32, device=dev)
            slot = torch.empty((n,), dtype=torch.int32, device=dev)
            self.be.bucket(g32, N, cap, send, slot, self._cnt, self._ovf)
            slot_t = torch.where(slot >= 0, slot, torch.full_like(slot, trash)).to(torch.int64)
            req = self._a2a(send)
            req_loc = torch.empty_like(req)
            self.be.localize(req, N, req_loc)
        else:                                                              # torch plan (CPU tests)
            slot, ov = bucket_slots(torch.where(g >= 0, g % N, g), N, cap)
            self.overflow |= ov
            slot_t = torch.where(slot >= 0, slot, torch.full_like(slot, trash))
            send = torch.full((trash + 1,), -1, dtype=torch.int32, device=dev)
            send.index_copy_(0, slot_t, g.to(torch.int32))
            req = self._a2a(send[:trash].contiguous()).to(torch.int64)
            req_loc = torch.where(req >= 0, torch.div(req, N, rounding_mode="floor"), req).to(torch.int32).contiguous()
        # ---- 2. owners gather, rows travel back.  Padding slots (request id -1) are skipped by the gather and
        # never selected below, so the exchange buffers need no clearing; slot `trash` is a zero row that
        # stands in for a lookup dropped by a full bucket.
        buf = self._buffers(n, trash, d)
        rows_out, rows_in, emb_grads, send_g = buf["rows_out"], buf["rows_in"], buf["emb_grads"], buf["send_g"]
        self.be.gather_rows(self.emb, req_loc, rows_out)
        self._a2a_into(rows_in[:trash], rows_out)
        indirect = hasattr(self.be, "grads_indirect") and self.be.direct_ok() and not self.copying
        if indirect:
            # ---- 3. local forward + backward on the rows where they arrived: lookup (b, f) reads row slot[b, f] of rows_in and its
            # gradient goes to row slot[b, f] of send_g (no reordering passes over the B x n_emb x d block)
            idx = buf.get("idx")
            if idx is None or idx.shape[0] != B:
                idx = buf["idx"] = torch.full((B, nf + 1), -1, dtype=torch.int32, device=dev)
            idx[:, :nf] = slot_t.view(B, nf)
            self.be.grads_indirect(dense, rows_in, idx, label, B * N, send_g, self.loss_accum)
        else:
            emb_rows = rows_in.index_select(0, slot_t)                       # [B*nf, d] in lookup order
            # ---- 3. local forward + backward
            self.be.grads(dense, emb_rows, label, B * N, emb_grads, self.loss_accum)
        # ---- 4. dense gradients: one all-reduce, then the dense rule on every replica
        self.be.dense_pack(self.flat)
        self._allreduce(self.flat)
        self.be.dense_apply(self.flat)
        # ---- 5. embedding gradients back to the owners (padding slots carry garbage and are skipped: id -1)
        if not indirect:
            send_g.index_copy_(0, slot_t, emb_grads)
        g_in = self._a2a(send_g[:trash])
        self.be.apply_rows(self.emb, req_loc, g_in)
        return None

    def _buffers(self, n, trash, d):
        key = (n, trash)
        if getattr(self, "_buf_key", None) != key:
            dev, f32 = self.device, torch.float32
            self._buf = dict(rows_out=torch.zeros((trash, d), dtype=f32, device=dev),
                             rows_in=torch.zeros((trash + 1, d), dtype=f32, device=dev),
                             emb_grads=torch.zeros((n, d), dtype=f32, device=dev),
                             send_g=torch.zeros((trash + 1, d), dtype=f32, device=dev))
            self._buf_key = key
        return self._buf

    def _a2a_into(self, out, x):
        if self.a2a_fn is not None:
            self.a2a_fn(out, x)
        elif self.world == 1 and not self.force_collectives:
            out.copy_(x)
        else:
            dist.all_to_all_single(out, x, group=self.group)

    # ---- results ---------------------------------------------------------------
    def loss_sum(self):
        """sum over the steps so far of the global-batch loss"""
        if hasattr(self.be, "stream"):
            self.be.stream.synchronize()
        t = self.loss_accum.clone()
        self._allreduce(t)
        return float(t.item())

    def check(self):
        if hasattr(self.be, "stream"):
            self.be.stream.synchronize()
        if bool(self.overflow.item()) or int(self._ovf.item()) != 0:
            raise RuntimeError("sharded DLRM: an exchange bucket overflowed (raise `slack`)")
        if bool(self.bad_id.item()):
            raise IndexError("sharded DLRM: embedding id out of range")
        if hasattr(self.be, "check"):
            self.be.check()
