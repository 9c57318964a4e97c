"""The row-sharded step (openrec_amd/sharded.py) on CPU: world size 2 over gloo
(and world size 1 in-process) must reproduce the single-process oracle on the
global batch.  The compute building blocks are the oracle here (the HIP ones
are parity-tested on the GPU); what is under test is routing, bucketing and
the all-to-all plan -- index work, bit-exact."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_err
from oracle import numpy_oracle as orc


def _global_case(model, seed=0, NU=101, NI=157, B=96, D=16):
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    steps = []
    for s in range(3):
        u = rng.integers(0, NU, B).astype(np.int32); p = rng.integers(0, NI, B).astype(np.int32); n = rng.integers(0, NI, B).astype(np.int32)
        u[:9] = 5; n[9:13] = p[9:13]
        steps.append((u, p, n))
    return U, V, b, steps


def _run_rank(rank, world, port, model, optk, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_ref_backend import OracleBackend
    from openrec_amd import sharded
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    U, V, b, steps = _global_case(model)
    be = OracleBackend(optk, 0.05)
    eng = sharded.ShardedPairwise(model, optk, U.shape[0], V.shape[0], U.shape[1], lr=0.05, rank=rank, world=world,
                                  device=torch.device("cpu"), backend=be, slack=1.5)
    eng.U.w[:] = U[rank::world]; eng.V.w[:] = V[rank::world]; eng.b.w[:] = b[rank::world]
    B = steps[0][0].shape[0]; per = B // world
    for (u, p, n) in steps:
        sl = slice(rank * per, (rank + 1) * per)
        eng.step(torch.from_numpy(u[sl].copy()), torch.from_numpy(p[sl].copy()), torch.from_numpy(n[sl].copy()))
    eng.check()
    loss, l2 = eng.loss_sums()
    np.savez(out % rank, U=eng.U.w, V=eng.V.w, b=eng.b.w, loss=loss, l2=l2)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
def test_sharded_equals_single_process(tmp_path, world, model, optk):
    out = str(tmp_path / "r%d.npz")
    if world == 1:
        _run_rank(0, 1, 0, model, optk, out)
    else:
        mp.spawn(_run_rank, args=(world, _free_port(), model, optk, out), nprocs=world, join=True)
    U, V, b, steps = _global_case(model)
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
    tl = tl2 = 0.0
    for (u, p, n) in steps:
        if model == "bpr":
            l, l2 = orc.bpr_step(U, V, b, u, p, n, o)
        else:
            l, l2 = orc.ucml_step(U, V, b, u, p, n, o, do_censor=False)
        tl += float(l); tl2 += float(l2)
    for r in range(world):
        g = np.load(out % r)
        assert rel_err(g["U"], U[r::world]) < 1e-5
        assert rel_err(g["V"], V[r::world]) < 1e-5
        assert rel_err(g["b"], b[r::world]) < 1e-5
        assert abs(float(g["loss"]) - tl) < 1e-5 * abs(tl) and abs(float(g["l2"]) - tl2) < 1e-5 * abs(tl2)


def test_bucket_slots_is_a_stable_partition():
    from openrec_amd.sharded import bucket_slots
    rng = np.random.default_rng(0)
    dest = torch.from_numpy(rng.integers(-1, 4, 500))
    slot, ov = bucket_slots(dest, 4, 200)
    assert not bool(ov)
    s = slot.numpy(); d = dest.numpy()
    assert ((s >= 0) == (d >= 0)).all()
    assert len(set(s[s >= 0])) == (s >= 0).sum()                 # injective
    assert (s[s >= 0] // 200 == d[d >= 0]).all()                 # right bucket
    for k in range(4):                                            # order preserved inside a bucket
        idx = np.nonzero(d == k)[0]
        assert (np.diff(s[idx]) == 1).all() and s[idx[0]] == k * 200
    slot, ov = bucket_slots(dest, 4, 10)
    assert bool(ov) and ((slot.numpy() >= 0).sum() == 40)


def _run_rank_ksteps(rank, world, port, model, optk, overlap, out, dedup=None):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_ref_backend import FastOracleBackend, FlaggedOracleBackend
    from openrec_amd import sharded
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    U, V, b, steps = _global_case(model)
    # SGD on the GPU takes the flagged route (duplicate flags, user rows updated by the gradient kernel); every second SGD
    # case here does too
    be = (FlaggedOracleBackend if optk == "sgd" and (model == "bpr") == bool(overlap) else FastOracleBackend)(optk, 0.05)
    eng = sharded.ShardedPairwise(model, optk, U.shape[0], V.shape[0], U.shape[1], lr=0.05, rank=rank, world=world,
                                  device=torch.device("cpu"), backend=be, slack=1.5, dedup=dedup)
    assert eng.fast
    eng.U.w[:] = U[rank::world]; eng.V.w[:] = V[rank::world]; eng.b.w[:] = b[rank::world]
    B = steps[0][0].shape[0]; per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    uid, pid, nid = (torch.from_numpy(np.stack([st[c][sl] for st in steps]).copy()) for c in range(3))    # [K, per]
    eng.steps(uid, pid, nid, plan_chunk=2, overlap=overlap)          # chunks of 2 + 1 steps
    eng.check()
    assert int(eng._ovf.item()) == 0
    loss, l2 = eng.loss_sums()
    np.savez(out % rank, U=eng.U.w, V=eng.V.w, b=eng.b.w, loss=loss, l2=l2)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
def test_sharded_kstep_paths_equal_single_process(tmp_path, world, model, optk, overlap, dedup=None):
    """What `bench.py --gpus N` runs -- `ShardedPairwise.steps`: the exchange plan of a chunk of steps in one all-to-all per
    phase, and the overlapped form with two half-batches per step and ASYNCHRONOUS all-to-alls -- over real collectives with
    two ranks (gloo).  The device kernels of the plan are restated in tests/sharded_ref_backend.py (FastOracleBackend); the
    GPU tests hold the kernels themselves against the same oracle on a virtual cluster."""
    out = str(tmp_path / "k%d.npz")
    if world == 1:
        _run_rank_ksteps(0, 1, 0, model, optk, overlap, out, dedup)
    else:
        mp.spawn(_run_rank_ksteps, args=(world, _free_port(), model, optk, overlap, out, dedup), nprocs=world, join=True)
    U, V, b, steps = _global_case(model)
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
    tl = tl2 = 0.0
    for (u, p, n) in steps:
        l, l2 = orc.bpr_step(U, V, b, u, p, n, o) if model == "bpr" else orc.ucml_step(U, V, b, u, p, n, o, do_censor=False)
        tl += float(l); tl2 += float(l2)
    for r in range(world):
        g = np.load(out % r)
        assert rel_err(g["U"], U[r::world]) < 1e-5 and rel_err(g["V"], V[r::world]) < 1e-5 and rel_err(g["b"], b[r::world]) < 1e-5
        assert abs(float(g["loss"]) - tl) < 1e-5 * abs(tl) and abs(float(g["l2"]) - tl2) < 1e-5 * abs(tl2)


@pytest.mark.parametrize("model,optk,overlap", [("bpr", "sgd", True), ("bpr", "adagrad", False), ("ucml", "sgd", True)])
def test_sharded_kstep_paths_world4(tmp_path, model, optk, overlap):
    """four ranks: the [dest][step][slot] <-> [step][src][slot] transposes of the chunk-wide exchanges are their own inverse
    at world 2 -- here they are not"""
    test_sharded_kstep_paths_equal_single_process(tmp_path, 4, model, optk, overlap)


@pytest.mark.parametrize("dedup", [False, True])
@pytest.mark.parametrize("model,optk,overlap", [("bpr", "sgd", False), ("bpr", "adagrad", True), ("ucml", "adam", False)])
def test_sharded_kstep_paths_with_and_without_request_dedup(tmp_path, model, optk, overlap, dedup):
    """The request plan both ways, whatever the size heuristic (ShardedPairwise._dedup_for) would pick for this case: one slot
    per reference, and one slot per distinct item of a list with the owner's gradients summed per slot before they travel."""
    test_sharded_kstep_paths_equal_single_process(tmp_path, 2, model, optk, overlap, dedup)


def test_library_engine_bucket_capacities_equal_the_python_engine():
    """orx_sharded_caps (host arithmetic of the library's engine, no device needed) against ShardedPairwise._cap: the two engines
    must size their exchange buckets alike, or a job that mixes them (per-phase tests against the library's K-step call) disagrees
    about what overflows."""
    import ctypes
    import math
    from openrec_amd import _ffi
    lib = _ffi.load()
    for B in (1, 7, 4096, 65536, 32768):
        for world in (1, 2, 8, 64):
            for slack in (1.0, 1.05, 1.5):
                c1, c2 = ctypes.c_int64(), ctypes.c_int64()
                _ffi.check(lib.orx_sharded_caps(B, world, slack, ctypes.byref(c1), ctypes.byref(c2)))
                cap = lambda n: int(math.ceil(n / world * slack + 6 * math.sqrt(n / world) + 16))
                assert c1.value == cap(B), (B, world, slack)
                assert c2.value == cap(2 * world * cap(B)), (B, world, slack)


def _run_rank_ckpt(rank, world, port, optk, out, ckdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_ref_backend import OracleBackend
    from openrec_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    U, V, b, steps = _global_case("bpr")
    per = steps[0][0].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)

    def engine():
        e = sharded.ShardedPairwise("bpr", optk, U.shape[0], V.shape[0], U.shape[1], lr=0.05, rank=rank, world=world,
                                    device=torch.device("cpu"), backend=OracleBackend(optk, 0.05), slack=1.5)
        e.U.w[:] = U[rank::world]; e.V.w[:] = V[rank::world]; e.b.w[:] = b[rank::world]
        return e
    a = engine()
    for (u, p, n) in steps[:2]:
        a.step(torch.from_numpy(u[sl].copy()), torch.from_numpy(p[sl].copy()), torch.from_numpy(n[sl].copy()))
    a.save(ckdir)                                    # every rank writes its own shard into the same directory
    u, p, n = steps[2]
    a.step(torch.from_numpy(u[sl].copy()), torch.from_numpy(p[sl].copy()), torch.from_numpy(n[sl].copy()))
    r = engine()                                     # a fresh engine resumes from the directory
    r.U.w[:] = 0; r.V.w[:] = 0; r.b.w[:] = 0
    r.load(ckdir)
    r.step(torch.from_numpy(u[sl].copy()), torch.from_numpy(p[sl].copy()), torch.from_numpy(n[sl].copy()))
    np.savez(out % rank, U=a.U.w, V=a.V.w, b=a.b.w, U2=r.U.w, V2=r.V.w, b2=r.b.w)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
def test_sharded_checkpoint_resumes_at_world_2(tmp_path, optk):
    """ShardedPairwise.save / load: every rank writes its shard (tables, optimizer state) into one directory; a fresh engine of the
    same layout that loads it continues exactly like the uninterrupted run (gloo, two ranks, oracle compute)."""
    out = str(tmp_path / "c%d.npz")
    mp.spawn(_run_rank_ckpt, args=(2, _free_port(), optk, out, str(tmp_path / "ck")), nprocs=2, join=True)
    for r in range(2):
        g = np.load(out % r)
        for k in ("U", "V", "b"):
            assert np.array_equal(g[k], g[k + "2"]), (r, k)
    assert sorted(f for f in os.listdir(tmp_path / "ck")) == ["ref.rank0of2.pkl", "ref.rank1of2.pkl", "sharded_meta.rank0of2.json",
                                                                "sharded_meta.rank1of2.json"]


def _run_rank_hot_ckpt(rank, world, port, optk, out, ckdir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_ref_backend import OracleBackend
    from openrec_amd import sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    U, V, b, steps = _zipf_case("bpr", steps=4)
    per = steps[0][0].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)

    def engine(hot=64):
        e = sharded.ShardedPairwise("bpr", optk, U.shape[0], V.shape[0], U.shape[1], lr=0.05, rank=rank, world=world,
                                    device=torch.device("cpu"), backend=OracleBackend(optk, 0.05), slack=3.0, hot_items=hot)
        e.U.w[:] = U[rank::world]; e.V.w[:] = V[rank::world]; e.b.w[:] = b[rank::world]
        return e
    step = lambda e, k: e.step(*(torch.from_numpy(x[sl].copy()) for x in steps[k]))
    a = engine()
    step(a, 0); step(a, 1)
    a.save(ckdir)
    step(a, 2); step(a, 3)
    a.sync_hot()
    r = engine()                                     # a fresh engine resumes from the directory: tables AND the replicas' optimizer state
    r.U.w[:] = 0; r.V.w[:] = 0; r.b.w[:] = 0
    r.load(ckdir)
    step(r, 2); step(r, 3)
    r.sync_hot()
    refused = False
    try:                                             # another replica layout cannot take over the replicas' optimizer state
        engine(hot=32).load(ckdir)
    except ValueError:
        refused = True
    np.savez(out % rank, U=a.U.w, V=a.V.w, b=a.b.w, U2=r.U.w, V2=r.V.w, b2=r.b.w, refused=refused)
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("optk", ["sgd", "adagrad", "adam"])
def test_sharded_checkpoint_with_hot_items_resumes_at_world_2(tmp_path, optk):
    """the replicated hot rows are trained on the replica, so their Adagrad accumulators / Adam moments live in the
    replica's slots; save() writes the replica (tables + slots) beside the shards and load() brings it back -- a resumed run
    continues bit for bit like the uninterrupted one.  A checkpoint of another `hot_items` is refused unless the optimizer is SGD."""
    out = str(tmp_path / "c%d.npz")
    mp.spawn(_run_rank_hot_ckpt, args=(2, _free_port(), optk, out, str(tmp_path / "ck")), nprocs=2, join=True)
    for r in range(2):
        g = np.load(out % r)
        for k in ("U", "V", "b"):
            assert np.array_equal(g[k], g[k + "2"]), (r, k)
        assert bool(g["refused"]) == (optk != "sgd")


def _zipf_case(model, seed=1, NU=101, NI=400, B=96, D=16, steps=3):
    """item ids ~ Zipf(1.05) over a vocabulary sorted by popularity: the head of the distribution takes most references"""
    rng = np.random.default_rng(seed)
    U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32)
    V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
    b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
    w = 1.0 / np.arange(1, NI + 1) ** 1.05
    cdf = np.cumsum(w / w.sum())
    draw = lambda: np.minimum(np.searchsorted(cdf, rng.random(B)), NI - 1).astype(np.int32)
    out = []
    for s in range(steps):
        out.append((rng.integers(0, NU, B).astype(np.int32), draw(), draw()))
    return U, V, b, out


def _run_rank_hot(rank, world, port, model, optk, hot, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_ref_backend import OracleBackend
    from openrec_amd import sharded
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    U, V, b, steps = _zipf_case(model)
    eng = sharded.ShardedPairwise(model, optk, U.shape[0], V.shape[0], U.shape[1], lr=0.05, rank=rank, world=world,
                                  device=torch.device("cpu"), backend=OracleBackend(optk, 0.05), slack=3.0, hot_items=hot)
    eng.U.w[:] = U[rank::world]; eng.V.w[:] = V[rank::world]; eng.b.w[:] = b[rank::world]
    per = steps[0][0].shape[0] // world
    sl = slice(rank * per, (rank + 1) * per)
    wire = 0
    real_a2a = eng._a2a

    def counting(send, recv=None):                        # floats this rank puts on the wire in the row / gradient exchanges
        nonlocal wire
        if send.dtype == torch.float32:
            wire += int((send.abs().sum(-1) > 0).sum())
        return real_a2a(send, recv)
    eng._a2a = counting
    for (u, p, n) in steps:
        eng.step(torch.from_numpy(u[sl].copy()), torch.from_numpy(p[sl].copy()), torch.from_numpy(n[sl].copy()))
    eng.check()
    eng.sync_hot()                                        # the trained hot rows go back into the owners' shards
    loss, l2 = eng.loss_sums()
    np.savez(out % rank, U=eng.U.w, V=eng.V.w, b=eng.b.w, loss=loss, l2=l2, wire=wire)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


@pytest.mark.parametrize("model,optk", [("bpr", "sgd"), ("bpr", "adagrad"), ("ucml", "sgd"), ("bpr", "adam")])
def test_hot_item_replication_equals_single_process_on_zipf_ids(tmp_path, model, optk):
    """with Zipf(1.05) item ids the 64 most popular of 400 items are replicated on both ranks (references read the
    replica, gradients are summed over the ranks by one all-reduce per step).  Same tables and losses as the single-process oracle on
    the global batch -- and far fewer item rows on the wire than without replication."""
    U, V, b, steps = _zipf_case(model)
    res = {}
    for hot in (0, 64):
        out = str(tmp_path / ("h%d_" % hot)) + "r%d.npz"
        mp.spawn(_run_rank_hot, args=(2, _free_port(), model, optk, hot, out), nprocs=2, join=True)
        res[hot] = [np.load(out % r) for r in range(2)]
    o = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
    Uo, Vo, bo = U.copy(), V.copy(), b.copy()
    tl = 0.0
    for (u, p, n) in steps:
        l, _ = orc.bpr_step(Uo, Vo, bo, u, p, n, o) if model == "bpr" else orc.ucml_step(Uo, Vo, bo, u, p, n, o, do_censor=False)
        tl += float(l)
    tol = 5e-5 if optk == "adam" else 1e-5
    for hot in (0, 64):
        for r in range(2):
            g = res[hot][r]
            assert rel_err(g["U"], Uo[r::2]) < tol and rel_err(g["V"], Vo[r::2]) < tol and rel_err(g["b"], bo[r::2]) < tol, (hot, r)
            assert abs(float(g["loss"]) - tl) < 1e-5 * abs(tl)
    wire0, wire1 = (sum(int(g["wire"]) for g in res[h]) for h in (0, 64))
    assert wire1 < 0.55 * wire0, (wire0, wire1)           # most item references were to hot rows
