"""Hybrid-parallel DLRM step on the GPU: N engines in N threads of one process share the GPU, the
collectives are replaced by an in-process fake cluster (there is one GPU on the test box).  Compute is the
C ABI (orx_gather_rows, orx_dlrm_grads, orx_dlrm_dense_*, orx_apply_rows); the result must equal the
single-process oracle on the global batch."""
import threading

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

CFG = dict(m_spa=16, ln_emb=[1000, 37, 5000, 3, 250], ln_bot=[64, 16], ln_top=[128, 64, 1], dense_dim=13)


class _FakeCluster:
    def __init__(self, N):
        self.N, self.bar, self.slots, self.red = N, threading.Barrier(N), [None] * N, [None] * N

    def a2a(self, rank):
        import torch

        def fn(recv, send):
            torch.cuda.synchronize()
            self.slots[rank] = send
            self.bar.wait()
            chunk = send.shape[0] // self.N
            for src in range(self.N):
                recv[src * chunk:(src + 1) * chunk] = self.slots[src][rank * chunk:(rank + 1) * chunk]
            torch.cuda.synchronize()
            self.bar.wait()
        return fn

    def allreduce(self, rank):
        import torch

        def fn(x):
            torch.cuda.synchronize()
            self.red[rank] = x.clone()
            self.bar.wait()
            x.zero_()
            for src in range(self.N):                      # same order on every rank: identical replicas
                x += self.red[src]
            torch.cuda.synchronize()
            self.bar.wait()
        return fn


def _case(steps, B, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for s in range(steps):
        dense = rng.normal(size=(B, CFG["dense_dim"])).astype(np.float32)
        sparse = np.stack([rng.integers(0, n, B) for n in CFG["ln_emb"]], 1).astype(np.int32)
        sparse[:40, 2] = 11
        label = (rng.random(B) < 0.3).astype(np.float32)
        out.append((dense, sparse, label))
    return out


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("optk,compat,loss,m_spa", [("sgd", False, "bce", 16), ("adagrad", False, "mse", 16), ("sgd", True, "mse", 16), ("adam", False, "bce", 16),
                                                    ("sgd", False, "bce", 32), ("adagrad", False, "mse", 64), ("adam", False, "mse", 32), ("sgd", True, "mse", 32)])
def test_virtual_cluster_dlrm_matches_oracle(world, optk, compat, loss, m_spa):
    """m_spa 32 / 64 (without reference_compat): the local step reads the exchanged rows in place and writes the row gradients
    straight into the buffer that travels back (orx_dlrm_grads_indirect); m_spa 16 and reference_compat: the copying form."""
    CFG = dict(globals()["CFG"], m_spa=m_spa, ln_bot=[64, m_spa])          # (dlrm.py: the bottom MLP ends at the embedding width)
    import torch
    from openrec_amd.sharded_dlrm import ShardedDLRM
    from oracle.dlrm_oracle import DLRMOracle
    from oracle import numpy_oracle as orc
    torch.cuda.init()                                   # in the main thread, before the rank threads touch the device
    dev = torch.device("cuda", 0)
    Bg, steps = 512, 3
    kw = dict(reference_compat=compat, loss_func=loss)
    ref = DLRMOracle(seed=5, **dict(CFG, **kw))
    comb0 = np.concatenate(ref.emb)
    cl = _FakeCluster(world)
    engs, errs = [None] * world, []
    data = _case(steps, Bg)

    def run(rank):
        try:
            e = ShardedDLRM(rank=rank, world=world, device=dev, opt=optk, lr=0.05, slack=2.0, seed=5,
                            a2a_fn=cl.a2a(rank), allreduce_fn=cl.allreduce(rank), **CFG, **kw)
            e.load_embeddings(comb0)
            # replicas start from the oracle's dense parameters
            for name, layers in (("bot", ref.bot), ("top", ref.top)):
                for l, (W, b) in enumerate(layers):
                    e.be.dense_param(name + "_w", l).write(W); e.be.dense_param(name + "_b", l).write(b.reshape(1, -1))
            engs[rank] = e
            assert e.be.direct_ok() == (m_spa >= 32 and not compat)
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            for dense, sparse, label in data:
                e.step(torch.from_numpy(dense[sl].copy()).to(dev), torch.from_numpy(sparse[sl].copy()).to(dev),
                       torch.from_numpy(label[sl].copy()).to(dev))
            torch.cuda.synchronize()
        except Exception as ex:                             # pragma: no cover
            errs.append(ex)
            cl.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    opt = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
    total = 0.0
    for dense, sparse, label in data:
        total += float(ref.step(dense, sparse, label, opt))
    comb = np.concatenate(ref.emb)
    tol = 5e-5 if optk == "adam" else 2e-5
    got = 0.0
    for r, e in enumerate(engs):
        e.check()
        assert rel_err(e.local_embeddings(), comb[r::world]) < tol
        for name, layers in (("bot", ref.bot), ("top", ref.top)):
            for l, (W, b) in enumerate(layers):
                assert rel_err(e.be.dense_param(name + "_w", l).read(), W) < tol, (name, l)
                assert rel_err(e.be.dense_param(name + "_b", l).read().reshape(-1), b) < tol, (name, l)
        got += float(e.loss_accum.item())
    assert abs(got - total) < 2e-5 * abs(total)


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("optk,loss,m_spa", [("sgd", "bce", 32), ("adagrad", "mse", 64), ("adam", "mse", 32)])
def test_library_dlrm_engine_matches_oracle(world, optk, loss, m_spa):
    """orx_sharded_dlrm_steps (the whole hybrid-parallel K-step loop inside the library: request / row / gradient exchanges, one
    all-reduce of the packed dense gradients, applies) with `world` ranks in threads of this process exchanging through an
    orx_vgroup (world 1: a one-rank communicator without RCCL), against the single-process oracle on the global batches --
    the RCCL side of the same schedule runs in tests/test_gpu_rccl_rank1.py / test_gpu_rccl_multirank.py."""
    import ctypes
    CFG = dict(globals()["CFG"], m_spa=m_spa, ln_bot=[64, m_spa])
    import torch
    from openrec_amd import _ffi
    from openrec_amd.sharded_dlrm import ShardedDLRM
    from oracle.dlrm_oracle import DLRMOracle
    from oracle import numpy_oracle as orc
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    Bg, steps = 512, 4
    kw = dict(reference_compat=False, loss_func=loss)
    ref = DLRMOracle(seed=5, **dict(CFG, **kw))
    comb0 = np.concatenate(ref.emb)
    lib = _ffi.load()
    vg = ctypes.c_void_p()
    if world > 1:
        _ffi.check(lib.orx_vgroup_create(world, ctypes.byref(vg)))
    engs, errs = [None] * world, []
    data = _case(steps, Bg)

    def run(rank):
        try:
            e = ShardedDLRM(rank=rank, world=world, device=dev, opt=optk, lr=0.05, slack=2.0, seed=5, vgroup=vg if world > 1 else None, **CFG, **kw)
            e.load_embeddings(comb0)
            for name, layers in (("bot", ref.bot), ("top", ref.top)):
                for l, (W, b) in enumerate(layers):
                    e.be.dense_param(name + "_w", l).write(W); e.be.dense_param(name + "_b", l).write(b.reshape(1, -1))
            engs[rank] = e
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            de = torch.from_numpy(np.stack([d[sl] for d, _, _ in data])).to(dev)
            sp = torch.from_numpy(np.stack([s_[sl] for _, s_, _ in data])).to(dev)
            la = torch.from_numpy(np.stack([y[sl] for _, _, y in data])).to(dev)
            torch.cuda.synchronize()
            e.steps(de[:3], sp[:3], la[:3])               # three steps in one library call ...
            assert e._comm is not None
            e.steps(de[3:], sp[3:], la[3:])               # ... and one more call
            e.be.stream.synchronize()
        except Exception as ex:                             # pragma: no cover
            errs.append(ex)
            if world > 1:
                lib.orx_vgroup_abort(vg)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    opt = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
    total = sum(float(ref.step(dense, sparse, label, opt)) for dense, sparse, label in data)
    comb = np.concatenate(ref.emb)
    tol = 5e-5 if optk == "adam" else 2e-5
    got = 0.0
    for r, e in enumerate(engs):
        e.check()
        assert rel_err(e.local_embeddings(), comb[r::world]) < tol
        for name, layers in (("bot", ref.bot), ("top", ref.top)):
            for l, (W, b) in enumerate(layers):
                assert rel_err(e.be.dense_param(name + "_w", l).read(), W) < tol, (name, l)
                assert rel_err(e.be.dense_param(name + "_b", l).read().reshape(-1), b) < tol, (name, l)
        got += float(e.loss_accum.item())
    assert abs(got - total) < 2e-5 * abs(total)
    engs.clear()
    import gc; gc.collect()
    if world > 1:
        lib.orx_vgroup_destroy(vg)
