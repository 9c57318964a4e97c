"""GPU parity of the sharded building blocks (orx_gather_rows / orx_pair_grads /
orx_apply_rows) and of the whole sharded step at world size 1 against the oracle."""
import os

import numpy as np
import pytest

from conftest import TOL, TOL_ADAM, rel_err

pytestmark = pytest.mark.gpu
LR_ADAM = 0.002
TOL = 1e-5


def _case(seed, NU, NI, B, D):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    u = rng.integers(0, NU, B).astype(np.int32); p = rng.integers(0, NI, B).astype(np.int32); n = rng.integers(0, NI, B).astype(np.int32)
    u[:11] = 2; n[11:19] = p[11:19]
    return U, V, b, u, p, n


@pytest.mark.parametrize("model", ["bpr", "ucml"])
@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("D", [50, 64, 128])
def test_sharded_world1_matches_oracle(model, optk, D):
    import torch
    from openrec_amd import sharded
    from oracle import numpy_oracle as orc
    torch.cuda.init()                                   # in the main thread, before the rank threads touch the device
    dev = torch.device("cuda", 0)
    U, V, b, u, p, n = _case(3, 700, 900, 2051, D)
    eng = sharded.ShardedPairwise(model, optk, 700, 900, D, lr=LR_ADAM if optk == "adam" else 0.05, rank=0, world=1, device=dev, slack=1.0)
    eng.U.write(U); eng.V.write(V); eng.b.write(b)
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(LR_ADAM)}[optk]()
    tl = tl2 = 0.0
    for s in range(3):
        uu, pp, nn = np.roll(u, s), np.roll(p, 5 * s), np.roll(n, 2 * s)
        eng.step(torch.from_numpy(uu).to(dev), torch.from_numpy(pp).to(dev), torch.from_numpy(nn).to(dev))
        if model == "bpr":
            l, l2 = orc.bpr_step(U, V, b, uu, pp, nn, o)
        else:
            l, l2 = orc.ucml_step(U, V, b, uu, pp, nn, o, do_censor=False)
        tl += float(l); tl2 += float(l2)
    eng.check()
    loss, l2s = eng.loss_sums()
    tol = TOL_ADAM if optk == "adam" else TOL       # (conftest.TOL_ADAM)
    assert abs(loss - tl) <= TOL * abs(tl) and abs(l2s - tl2) <= TOL * abs(tl2)
    assert rel_err(eng.U.read(), U) < tol and rel_err(eng.V.read(), V) < tol and rel_err(eng.b.read(), b) < tol


def test_gather_rows_skips_padding_and_is_bit_exact():
    import torch
    from openrec_amd import runtime as rt, _ffi
    torch.cuda.init()                                   # in the main thread, before the rank threads touch the device
    dev = torch.device("cuda", 0)
    ctx = rt.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(0)
    W = rng.normal(size=(300, 64)).astype(np.float32); bb = rng.normal(size=(300, 1)).astype(np.float32)
    t = rt.Table(300, 64, ctx).write(W); tb = rt.Table(300, 1, ctx).write(bb)
    ids = rng.integers(-1, 300, 1000).astype(np.int32)
    out = torch.full((1000, 68), 7.0, device=dev)
    _ffi.check(ctx._lib.orx_gather_rows(ctx._h, t._h, tb._h, torch.from_numpy(ids).to(dev).data_ptr(), 1000, out.data_ptr(), 68))
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    m = ids >= 0
    assert np.array_equal(o[m, :64], W[ids[m]]) and np.array_equal(o[m, 64], bb[ids[m], 0])
    assert (o[~m] == 7.0).all() and (o[:, 65:] == 7.0).all()


class _FakeCluster:
    """All-to-all between `N` engines living in N threads of this process (one GPU):
    exercises the device-side routing kernels with world > 1 without a second GPU."""

    def __init__(self, N):
        import threading
        self.N, self.bar, self.slots = N, threading.Barrier(N), [None] * N

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


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
@pytest.mark.parametrize("fast", [True, False, "planned", "overlapped"])
def test_virtual_cluster_matches_oracle(world, model, optk, fast):
    overlapped = fast == "overlapped"       # ... with every step cut into two half-batches (exchange / kernel overlap on RCCL)
    planned = fast in ("planned", "overlapped")     # K-step call: the exchange plan of all steps in one all-to-all per phase
    fast = bool(fast)
    import threading
    import torch
    from openrec_amd import sharded
    from oracle import numpy_oracle as orc
    torch.cuda.init()                                   # in the main thread, before the rank threads touch the device
    dev = torch.device("cuda", 0)
    NU, NI, D, Bg = 1001, 1503, 64, 4096
    U, V, b, u, p, n = _case(7, NU, NI, Bg, D)
    cl = _FakeCluster(world)
    engs, errs = [None] * world, []

    def run(rank):
        try:
            e = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=LR_ADAM if optk == "adam" else 0.05, rank=rank, world=world, device=dev,
                                        slack=1.5, a2a_fn=cl.a2a(rank), fast=fast)
            assert e.fast == fast
            e.U.write(U[rank::world]); e.V.write(V[rank::world]); e.b.write(b[rank::world])
            engs[rank] = e
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            if planned:
                stack = lambda f: torch.from_numpy(np.stack([f(s)[sl] for s in range(3)])).to(dev)
                e.steps(stack(lambda s: np.roll(u, s)), stack(lambda s: np.roll(p, 5 * s)), stack(lambda s: np.roll(n, 2 * s)),
                        plan_chunk=2, overlap=overlapped)
            for s in range(0 if not planned else 3, 3):
                uu, pp, nn = np.roll(u, s), np.roll(p, 5 * s), np.roll(n, 2 * s)
                e.step(torch.from_numpy(uu[sl].copy()).to(dev), torch.from_numpy(pp[sl].copy()).to(dev),
                       torch.from_numpy(nn[sl].copy()).to(dev))
            torch.cuda.synchronize()
        except Exception as ex:                         # pragma: no cover
            errs.append(ex)
            cl.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(LR_ADAM)}[optk]()
    tl = 0.0
    for s in range(3):
        uu, pp, nn = np.roll(u, s), np.roll(p, 5 * s), np.roll(n, 2 * s)
        l, _ = (orc.bpr_step if model == "bpr" else (lambda *a: orc.ucml_step(*a, do_censor=False)))(U, V, b, uu, pp, nn, o)
        tl += float(l)
    got = 0.0
    for r, e in enumerate(engs):
        assert int(e._ovf[0]) == 0 if e._ovf is not None else True
        assert not bool(e.overflow)
        tol = TOL_ADAM if optk == "adam" else TOL
        assert rel_err(e.U.read()[:len(U[r::world])], U[r::world]) < tol
        assert rel_err(e.V.read()[:len(V[r::world])], V[r::world]) < tol
        assert rel_err(e.b.read()[:len(b[r::world])], b[r::world]) < tol
        got += float(e.accum[0])
    assert abs(got - tl) <= 1e-5 * abs(tl)


@pytest.mark.parametrize("rows,D,n,K", [(40000, 128, 600, 15), (50, 64, 600, 40), (200000, 32, 3000, 60), (7, 8, 900, 25)])
def test_gather_apply_rows_adam_is_the_dense_decay_rule(rows, D, n, K):
    """The building blocks a host drives Adam with (orx_gather_rows / orx_opt_set_step / orx_apply_rows): the table
    is updated lazily (rows replay their gradient-free steps when gathered or given a gradient), every gather must
    return the rows of the dense every-row-every-step rule, and so must the table after K steps (full flush)."""
    import torch
    from openrec_amd import runtime as rt, _ffi
    from oracle import numpy_oracle as orc
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    ctx = rt.default_context(); lib = ctx._lib
    rng = np.random.default_rng(1)
    W0 = rng.uniform(-.05, .05, (rows, D)).astype(np.float32)
    t = rt.Table(rows, D, ctx).write(W0)
    opt = rt.Optimizer.adam(0.002, ctx=ctx)
    W = W0.astype(np.float64); oo = orc.AdamTFSparse(0.002)
    ids_all = rng.integers(0, rows, (K, n)).astype(np.int32); ids_all[:, :5] = -1          # padding slots
    g_all = rng.normal(0, 1e-3, (K, n, D)).astype(np.float32)
    out = torch.zeros((n, D), device=dev)
    for s in range(K):
        ids = torch.from_numpy(ids_all[s]).to(dev); g = torch.from_numpy(g_all[s]).to(dev)
        torch.cuda.synchronize()
        live = ids_all[s] >= 0
        _ffi.check(lib.orx_gather_rows(ctx._h, t._h, None, ids.data_ptr(), n, out.data_ptr(), D))
        ctx.synchronize()
        assert np.abs(out.cpu().numpy()[live] - W[ids_all[s][live]]).max() <= 2e-5 * 0.05, s
        opt.advance([t])
        _ffi.check(lib.orx_apply_rows(ctx._h, opt._h, t._h, None, ids.data_ptr(), n, g.data_ptr(), D))
        ctx.synchronize()
        oo.begin_step(); oo.apply(W, ids_all[s][live], g_all[s][live].astype(np.float64), key="W")
    assert rel_err(t.read(), W) < 2e-5
    assert rel_err(opt.slot(t, 0), oo.m["W"]) < 5e-5 and rel_err(opt.slot(t, 1), oo.v["W"]) < 5e-4


@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
def test_sharded_engine_checkpoint_resumes(tmp_path, optk):
    """ShardedPairwise.save / load on the device backend (one rank): tables, optimizer slots and Adam's step counter of the shard
    go to `table.<name>.rank0of1.npy` ... and a fresh engine that loads them continues like the uninterrupted run."""
    import torch
    from openrec_amd import sharded
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    NU, NI, D, B, K = 900, 1100, 64, 1024, 6
    ids = [torch.from_numpy(rng.integers(0, n, (K, B)).astype(np.int32)).to(dev) for n in (NU, NI, NI)]
    lr = 0.002 if optk == "adam" else 0.05

    def engine():
        return sharded.ShardedPairwise("bpr", optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, seed=3)
    a = engine()
    a.steps(ids[0][:3], ids[1][:3], ids[2][:3])
    a.save(str(tmp_path / "ck"))
    assert "table.U.rank0of1.npy" in os.listdir(tmp_path / "ck") and "manifest.rank0of1.json" in os.listdir(tmp_path / "ck")
    a.steps(ids[0][3:], ids[1][3:], ids[2][3:])
    a.check()
    r = engine()
    if optk != "sgd":
        r.steps(ids[0][:1], ids[1][:1], ids[2][:1])          # (slots exist once a step has used them)
    r.load(str(tmp_path / "ck"))
    r.steps(ids[0][3:], ids[1][3:], ids[2][3:])
    r.check()
    for x, y in ((a.U, r.U), (a.V, r.V), (a.b, r.b)):
        assert np.abs(x.read() - y.read()).max() <= 1e-6 * np.abs(x.read()).max()


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
def test_hot_item_replication_on_the_device_backend(world, model, optk):
    """ShardedPairwise(hot_items=H) with the HIP building blocks (orx_gather_rows / orx_pair_grads / orx_apply_rows on shards AND
    replicas): Zipf(1.05) item ids, the 256 most popular of 3000 items replicated, `world` engines in threads of this process
    (in-process all-to-all / all-reduce), against the single-process oracle on the global batch."""
    import threading
    import torch
    from openrec_amd import sharded
    from oracle import numpy_oracle as orc
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    NU, NI, D, Bg, H, steps = 1001, 3000, 64, 2048, 256, 3
    rng = np.random.default_rng(9)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    draw = lambda: np.minimum(np.searchsorted(cdf, rng.random(Bg)), NI - 1).astype(np.int32)
    data = [(rng.integers(0, NU, Bg).astype(np.int32), draw(), draw()) for _ in range(steps)]
    assert np.mean(np.concatenate([d[1] for d in data]) < H) > 0.5          # most references are to replicated rows
    cl = _FakeCluster(world)
    red = [None] * world

    def allreduce(rank):
        def fn(x):
            torch.cuda.synchronize()
            red[rank] = x.clone()
            cl.bar.wait()
            x.zero_()
            for src in range(world):
                x += red[src]
            torch.cuda.synchronize()
            cl.bar.wait()
        return fn
    lr = LR_ADAM if optk == "adam" else 0.05
    engs, errs = [None] * world, []

    def run(rank):
        try:
            e = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=rank, world=world, device=dev, slack=3.0, hot_items=H,
                                        a2a_fn=cl.a2a(rank) if world > 1 else None, allreduce_fn=allreduce(rank) if world > 1 else None)
            e.U.write(U[rank::world]); e.V.write(V[rank::world]); e.b.write(b[rank::world])
            engs[rank] = e
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            for (u, p, n) in data:
                e.step(torch.from_numpy(u[sl].copy()).to(dev), torch.from_numpy(p[sl].copy()).to(dev), torch.from_numpy(n[sl].copy()).to(dev))
            e.check()
            e.sync_hot()
        except Exception as ex:                             # pragma: no cover
            errs.append(ex)
            cl.bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(LR_ADAM)}[optk]()
    tl = 0.0
    for (u, p, n) in data:
        l, _ = orc.bpr_step(U, V, b, u, p, n, o) if model == "bpr" else orc.ucml_step(U, V, b, u, p, n, o, do_censor=False)
        tl += float(l)
    tol = TOL_ADAM if optk == "adam" else 2e-5
    got = 0.0
    for r, e in enumerate(engs):
        for have, want, nm in ((e.U.read(), U, "U"), (e.V.read(), V, "V"), (e.b.read(), b, "b")):
            assert rel_err(have[:len(want[r::world])], want[r::world]) < tol, (r, nm)
        got += float(e.accum[0])
    assert abs(got - tl) <= 2e-5 * abs(tl)
