"""The sharded step inside the library (openrec_amd/csrc/sharded_engine.hip): the pieces that the per-phase tests do not see.
  * orx_shard_regroup: the regrouping around the plan's exchanges, against a NumPy transpose;
  * orx_shard_request_dedup_steps: the per-destination dedup plan, field by field against its restatement in
    tests/sharded_ref_backend.py (the contract the gloo world-2 tests run on), including a bucket overflow;
  * orx_sharded_pairwise_steps on a one-rank communicator without RCCL: every flag combination against the oracle on a batch
    with heavy item duplication (shared slots summed) -- the RCCL side of it runs in tests/test_gpu_rccl_rank1.py."""
import ctypes

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_regroup_is_the_transpose_of_steps_and_peers():
    import torch
    from openrec_amd import runtime as rt, _ffi
    ctx = rt.default_context(); lib = ctx._lib
    dev = torch.device("cuda", 0)
    for K, N, w in ((5, 4, 7), (1, 8, 3), (64, 2, 300), (3, 1, 11)):
        x = torch.arange(K * N * w, dtype=torch.int32, device=dev).reshape(K, N, w)
        y = torch.empty((N, K, w), dtype=torch.int32, device=dev); z = torch.empty_like(x)
        torch.cuda.synchronize()
        _ffi.check(lib.orx_shard_regroup(ctx._h, x.data_ptr(), y.data_ptr(), K, N, w, 0))
        _ffi.check(lib.orx_shard_regroup(ctx._h, y.data_ptr(), z.data_ptr(), K, N, w, 1))
        ctx.synchronize()
        assert torch.equal(y, x.transpose(0, 1).contiguous()) and torch.equal(z, x)


@pytest.mark.parametrize("world,T,NI,cap", [(1, 700, 90, 1500), (4, 1000, 333, 600), (8, 513, 5000, 200), (2, 800, 40, 10)])
def test_dedup_request_plan_matches_its_restatement(world, T, NI, cap):
    import torch
    from openrec_amd import runtime as rt, _ffi
    from sharded_ref_backend import FastOracleBackend
    rng = np.random.default_rng(world * 1000 + T)
    K = 3
    trip = np.stack([rng.integers(0, 5000, (K, T)), rng.integers(0, NI, (K, T)), rng.integers(0, NI, (K, T))], -1).astype(np.int32)
    trip[:, rng.random(T) < 0.1] = -1                                   # empty slots of the receive buffer
    trip[0, :40, 1] = 7; trip[0, 40:60, 2] = 7                          # one item asked for 60 times (p and n references)
    ctx = rt.default_context(); lib = ctx._lib
    dev = torch.device("cuda", 0)
    d = dict(send=torch.empty((K, world * cap), dtype=torch.int32, device=dev), slot=torch.empty((K, 2 * T), dtype=torch.int32, device=dev),
             u_loc=torch.empty((K, T), dtype=torch.int32, device=dev), dup=torch.empty((K, 2 * T), dtype=torch.uint8, device=dev),
             ovf=torch.zeros(1, dtype=torch.int32, device=dev), so=torch.empty((K, 2 * T, 2), dtype=torch.int32, device=dev),
             sl=torch.empty((K, T, 2), dtype=torch.int32, device=dev), sc=torch.empty(K, dtype=torch.int32, device=dev))
    tt = torch.from_numpy(trip).to(dev)
    torch.cuda.synchronize()
    _ffi.check(lib.orx_shard_request_dedup_steps(ctx._h, tt.data_ptr(), K, T, world, cap, NI, d["send"].data_ptr(), d["slot"].data_ptr(),
                                                 d["u_loc"].data_ptr(), d["dup"].data_ptr(), d["so"].data_ptr(), d["sl"].data_ptr(), d["sc"].data_ptr(), d["ovf"].data_ptr()))
    ctx.synchronize()
    r = dict(send=torch.empty((K, world * cap), dtype=torch.int32), slot=torch.empty((K, 2 * T), dtype=torch.int32),
             u_loc=torch.empty((K, T), dtype=torch.int32), dup=torch.empty((K, 2 * T), dtype=torch.uint8), ovf=torch.zeros(1, dtype=torch.int32))
    FastOracleBackend("sgd", 0.1).shard_request_dedup_steps(torch.from_numpy(trip), world, cap, NI, r["send"], r["slot"], r["u_loc"], r["dup"], r["ovf"])
    for k in ("send", "slot", "u_loc", "dup", "ovf"):
        assert torch.equal(d[k].cpu(), r[k]), k
    assert int(r["ovf"]) == (1 if cap == 10 else 0)
    # the list of shared slots: every slot with more than one reference, once
    for k in range(K):
        sl_k, dup_k = d["slot"].cpu().numpy()[k], d["dup"].cpu().numpy()[k] != 0
        want = set(int(x) for x in sl_k[dup_k & (sl_k >= 0)])
        got = [int(x) for x in d["sl"].cpu().numpy()[k, :int(d["sc"][k]), 0]]
        assert len(got) == len(set(got)) and set(got) == want
    s, live = d["slot"].cpu().numpy()[0], trip[0, :, 0] >= 0
    if cap != 10:
        shared = set(s[:40][live[:40]]) | set(s[T + 40:T + 60][live[40:60]])
        assert len(shared) == 1 and -1 not in shared and d["dup"].cpu().numpy()[0][:40][live[:40]].all()     # the 60 references share one slot


@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
@pytest.mark.parametrize("dedup", [True, False])
def test_library_engine_with_and_without_dedup(optk, dedup):
    import torch
    from openrec_amd import sharded
    from oracle import numpy_oracle as orc
    rng = np.random.default_rng(3)
    NU, NI, D, B, K = 500, 60, 32, 2048, 5                              # 4096 item references over 60 items: every slot is shared
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32); nid = rng.integers(0, NI, (K, B)).astype(np.int32)
    dev = torch.device("cuda", 0)
    lr = 0.002 if optk == "adam" else 0.05
    eng = sharded.ShardedPairwise("bpr", optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=1.0, dedup=dedup)
    eng.U.write(U); eng.V.write(V); eng.b.write(b)
    tu, tp, tn = (torch.from_numpy(x).to(dev) for x in (uid, pid, nid))
    torch.cuda.synchronize()
    eng.steps(tu, tp, tn, plan_chunk=2)
    assert eng._comm is not None                                        # the K-step call ran inside the library
    eng.check()
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
    tl = sum(float(orc.bpr_step(U, V, b, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
    loss, _ = eng.loss_sums()
    assert abs(loss - tl) <= 1e-5 * abs(tl)
    tol = 2e-4 if optk == "adam" else 2e-5                               # (sums of ~70 gradient rows per item row in fp32)
    for got, want, nm in ((eng.U.read(), U, "U"), (eng.V.read(), V, "V"), (eng.b.read(), b, "b")):
        assert rel_err(got, want) < tol, (nm, rel_err(got, want))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
@pytest.mark.parametrize("overlap,dedup", [(False, False), (True, True), (False, True), (True, False)])
def test_library_engine_with_virtual_ranks(world, model, optk, overlap, dedup):
    """orx_sharded_pairwise_steps with world > 1 on one GPU: `world` ranks in threads of this process, each with its own context,
    exchanging through an orx_vgroup (host barrier + copy kernels in place of the ncclSend / ncclRecv groups).  The whole schedule
    of the engine -- plan regrouped by peer, gathers, gradients, applies, the halves of the overlapped path -- against the
    single-process oracle on the global batch."""
    import threading
    import torch
    from openrec_amd import sharded, _ffi
    from oracle import numpy_oracle as orc
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(7)
    NU, NI, D, Bg, K = 1001, 1503, 64, 4096, 5
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    uid = rng.integers(0, NU, (K, Bg)).astype(np.int32); pid = rng.integers(0, NI, (K, Bg)).astype(np.int32); nid = rng.integers(0, NI, (K, Bg)).astype(np.int32)
    uid[:, :17] = 5; pid[:, :40] = 9                                    # hot rows: duplicates within and across the ranks' slices
    lib = _ffi.load()
    vg = ctypes.c_void_p()
    _ffi.check(lib.orx_vgroup_create(world, ctypes.byref(vg)))
    lr = 0.002 if optk == "adam" else 0.05
    engs, errs = [None] * world, []

    def run(rank):
        try:
            e = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=rank, world=world, device=dev, slack=1.5, vgroup=vg, dedup=dedup)
            e.U.write(U[rank::world]); e.V.write(V[rank::world]); e.b.write(b[rank::world])
            engs[rank] = e
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            tu, tp, tn = (torch.from_numpy(np.ascontiguousarray(x[:, sl])).to(dev) for x in (uid, pid, nid))
            torch.cuda.synchronize()
            e.steps(tu, tp, tn, plan_chunk=2, overlap=overlap)
            assert e._comm is not None
            e.be.stream.synchronize()
        except Exception as ex:                                         # pragma: no cover
            errs.append(ex)
            lib.orx_vgroup_abort(vg)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
    step = orc.bpr_step if model == "bpr" else (lambda *a: orc.ucml_step(*a, do_censor=False))
    tl = sum(float(step(U, V, b, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
    got = 0.0
    tol = 5e-5 if optk == "adam" else 1e-5
    for r, e in enumerate(engs):
        assert int(e._ovf[0]) == 0
        for have, want, nm in ((e.U.read(), U, "U"), (e.V.read(), V, "V"), (e.b.read(), b, "b")):
            assert rel_err(have[:len(want[r::world])], want[r::world]) < tol, (r, nm)
        got += float(e.accum[0])
    assert abs(got - tl) <= 1e-5 * abs(tl)
    engs.clear()
    import gc; gc.collect()
    lib.orx_vgroup_destroy(vg)


@pytest.mark.parametrize("world", [1, 2, 4])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
@pytest.mark.parametrize("overlap", [False, True])
def test_library_engine_replicates_hot_items(world, model, optk, overlap):
    """orx_sharded_pairwise_steps_hot: with Zipf(1.05) item ids the H most popular items are
    replicated on every rank -- their references read the local replica, ask nobody and send nothing; their gradients are summed
    per item, then over the ranks by ONE all-reduce per step, and applied to every replica alike.  Virtual ranks in threads of this
    process (world > 1) or a one-rank communicator, against the single-process oracle on the global batch; the replicas of all
    ranks must be bit-identical, and the item rows that still travel must be far fewer than without replication."""
    import threading
    import torch
    from openrec_amd import sharded, _ffi
    from oracle import numpy_oracle as orc
    torch.cuda.init()
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    NU, NI, D, Bg, K, H = 1001, 1503, 64, 4096, 5, 128
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    draw = lambda: np.minimum(np.searchsorted(cdf, rng.random((K, Bg))), NI - 1).astype(np.int32)
    uid = rng.integers(0, NU, (K, Bg)).astype(np.int32); pid, nid = draw(), draw()
    share = float(((pid < H).sum() + (nid < H).sum()) / (2 * K * Bg))
    assert share > 0.5                                                  # most item references are hot
    lib = _ffi.load()
    vg = ctypes.c_void_p()
    if world > 1:
        _ffi.check(lib.orx_vgroup_create(world, ctypes.byref(vg)))
    # (lr * references of the hottest row < 1: SGD's l2 term moves a row referenced c times by lr * c * row per step)
    lr = 0.002 if optk == "adam" else 0.0005
    engs, errs = [None] * world, []

    def run(rank):
        try:
            e = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=rank, world=world, device=dev, slack=3.0, hot_items=H,
                                        vgroup=vg if world > 1 else None)
            e.U.write(U[rank::world]); e.V.write(V[rank::world]); e.b.write(b[rank::world])
            e.Vh.write(V[:H]); e.bh.write(b[:H]); e._hot_loaded = True     # (load_hot() is a torch.distributed collective: not among threads)
            engs[rank] = e
            per = Bg // world
            sl = slice(rank * per, (rank + 1) * per)
            tu, tp, tn = (torch.from_numpy(np.ascontiguousarray(x[:, sl])).to(dev) for x in (uid, pid, nid))
            torch.cuda.synchronize()
            if world > 1:
                e.steps(tu, tp, tn, plan_chunk=2, overlap=overlap)
            else:
                e.steps(tu, tp, tn, plan_chunk=2)
            assert e._comm is not None and e._fast_hot                  # the K-step call ran inside the library, replicas and all
            e.be.stream.synchronize()
        except Exception as ex:                                         # pragma: no cover
            errs.append(ex)
            if world > 1:
                lib.orx_vgroup_abort(vg)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
    step = orc.bpr_step if model == "bpr" else (lambda *a: orc.ucml_step(*a, do_censor=False))
    Uo, Vo, bo = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)      # (the hottest rows sum ~1000 gradients per step)
    tl = sum(float(step(Uo, Vo, bo, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
    got = 0.0
    tol = 5e-5 if optk == "adam" else 1e-5
    vh0, bh0 = engs[0].Vh.read(), engs[0].bh.read()
    for r, e in enumerate(engs):
        assert int(e._ovf[0]) == 0
        assert np.array_equal(e.Vh.read(), vh0) and np.array_equal(e.bh.read(), bh0), "replicas diverged"
        e.sync_hot()                                                    # the trained hot rows go back into the owners' shards
        for have, want, nm in ((e.U.read(), Uo, "U"), (e.V.read(), Vo, "V"), (e.b.read(), bo, "b")):
            assert rel_err(have[:len(want[r::world])], want[r::world]) < tol, (r, nm, rel_err(have[:len(want[r::world])], want[r::world]))
        got += float(e.accum[0])
    assert abs(got - tl) <= 1e-5 * abs(tl)
    assert rel_err(vh0[:H], Vo[:H]) < tol
    engs.clear()
    import gc; gc.collect()
    if world > 1:
        lib.orx_vgroup_destroy(vg)


def test_hot_items_shrink_the_exchanged_buckets():
    """the exchanged buckets travel whole (fixed capacities): with `hot_cold_fraction` they are sized for the references that are
    not replicated -- same results; a share set too small is an overflow that check() reports (which also shows that the buckets
    are sized by it: the wire bytes of an exchange are world - 1 buckets of that capacity, whatever they hold)"""
    import torch
    from openrec_amd import sharded
    from oracle import numpy_oracle as orc
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(12)
    NU, NI, D, B, K, H = 2000, 3000, 64, 4096, 4, 256
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    draw = lambda: np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1).astype(np.int32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid, nid = draw(), draw()
    cold = float(max(((pid[k] >= H).mean() + (nid[k] >= H).mean()) / 2 for k in range(K)))
    assert cold < 0.5
    lr = 0.0005
    res = {}
    for frac in (1.0, cold * 1.05 + 0.01, cold * 0.5):
        e = sharded.ShardedPairwise("bpr", "sgd", NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=1.05, hot_items=H, hot_cold_fraction=frac)
        e.U.write(U); e.V.write(V); e.b.write(b)
        tu, tp, tn = (torch.from_numpy(x).to(dev) for x in (uid, pid, nid))
        torch.cuda.synchronize()
        e.steps(tu, tp, tn, plan_chunk=2)
        if frac < cold:
            with pytest.raises(RuntimeError):
                e.check()
            continue
        e.check()
        e.sync_hot()
        res[frac] = (e.U.read(), e.V.read(), e.b.read())
    Uo, Vo, bo = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
    oo = orc.SGD(lr)
    for s in range(K):
        orc.bpr_step(Uo, Vo, bo, uid[s], pid[s], nid[s], oo)
    assert len(res) == 2                                             # (the third share overflowed, above: the buckets ARE sized by it)
    for got in res.values():
        assert rel_err(got[0], Uo) < 1e-5 and rel_err(got[1], Vo) < 1e-5 and rel_err(got[2], bo) < 1e-5


@pytest.mark.parametrize("optk", ["adagrad", "adam"])
def test_hot_replica_state_survives_a_checkpoint_on_the_device(tmp_path, optk):
    """ADVICE r4 on the device backend: the replicated rows are trained on the replica, so their Adagrad accumulators / Adam moments
    live in the replica's slots; ShardedPairwise.save writes the replica (tables + slots) beside the shards, load brings it back, and
    a resumed run continues like the uninterrupted one (library engine, one rank)."""
    import torch
    from openrec_amd import sharded
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(21)
    NU, NI, D, B, K, H = 1500, 2000, 64, 2048, 4, 128
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    draw = lambda: np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1).astype(np.int32)
    uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid, nid = draw(), draw()
    lr = 0.002 if optk == "adam" else 0.0005
    tu, tp, tn = (torch.from_numpy(x).to(dev) for x in (uid, pid, nid))

    def engine():
        e = sharded.ShardedPairwise("bpr", optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=3.0, hot_items=H)
        e.U.write(U); e.V.write(V); e.b.write(b)
        return e
    a = engine()
    a.steps(tu[:2], tp[:2], tn[:2], plan_chunk=2)
    a.save(str(tmp_path / "ck"))
    a.steps(tu[2:], tp[2:], tn[2:], plan_chunk=2)
    a.check(); a.sync_hot()
    r = engine()
    r.U.fill(0.0); r.V.fill(0.0); r.b.fill(0.0)
    r.load(str(tmp_path / "ck"))
    r.steps(tu[2:], tp[2:], tn[2:], plan_chunk=2)
    r.check(); r.sync_hot()
    assert a._fast_hot and r._fast_hot and r._comm is not None
    tol = 5e-5 if optk == "adam" else 1e-6
    for x, y, nm in ((a.U.read(), r.U.read(), "U"), (a.V.read(), r.V.read(), "V"), (a.b.read(), r.b.read(), "b"),
                     (a.Vh.read(), r.Vh.read(), "Vh"), (a.bh.read(), r.bh.read(), "bh")):
        assert rel_err(y, x) < tol, (nm, rel_err(y, x))
    with pytest.raises(ValueError):                  # another replica layout cannot take over the replicas' optimizer state
        sharded.ShardedPairwise("bpr", optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=3.0, hot_items=64).load(str(tmp_path / "ck"))
