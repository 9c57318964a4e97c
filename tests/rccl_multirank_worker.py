"""Worker of tests/test_gpu_rccl_multirank.py: ONE of W ranks (one per GPU) launched by torch.distributed.run with backend
"nccl" (= RCCL over xGMI), the way the driver launches `bench.py --gpus N`.  Every rank holds its row shard (row r on rank
r % W) and its slice of the global batch; after the run every rank's shard must equal the single-process NumPy oracle on
the GLOBAL batch (paths relative to /root/reference: recommenders/bpr.py:21-37, ucml.py:21-48, dlrm.py:63-100 are what the
oracle restates).  Covered: the library's K-step engine (`orx_sharded_pairwise_steps`: ncclSend / ncclRecv groups on the
context's stream, halves overlapped or sequential, request dedup on / off), the per-phase Python engine over
torch.distributed, SGD / Adagrad / lazily-applied Adam / UCML, and the hybrid-parallel DLRM step."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


PAIRWISE_CASES = (  # model, optimizer, dim, engine (None = library), overlap, dedup
    ("bpr", "sgd", 64, None, None, None), ("bpr", "sgd", 64, None, False, True), ("bpr", "sgd", 64, None, True, False),
    ("bpr", "sgd", 64, "python", None, None), ("bpr", "adagrad", 64, None, None, None), ("bpr", "adagrad", 64, "python", None, None),
    ("ucml", "sgd", 128, None, None, None), ("ucml", "sgd", 128, None, False, False),
    ("bpr", "adam", 64, None, None, None), ("bpr", "adam", 64, None, False, True), ("bpr", "adam", 64, "python", None, None))
HOT_CASES = (("bpr", "sgd", None), ("bpr", "adagrad", False), ("bpr", "adam", None))      # model, optimizer, overlap: hot-item replication in the library's engine
DLRM_CASES = (("sgd", "bce", None), ("adagrad", "bce", "python"), ("adam", "mse", None), ("sgd", "mse", "python"))      # optimizer, loss, engine (None = library)


def main():
    import torch
    import torch.distributed as dist
    from openrec_amd import sharded
    from openrec_amd.sharded_dlrm import ShardedDLRM
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == world

    def say(msg):
        if rank == 0:
            print(msg, flush=True)

    # ---- row-sharded BPR / UCML: 6 steps through the K-step path (chunks of 4 + 2), 2 through the per-step path
    for model, optk, D, engine, overlap, dedup in PAIRWISE_CASES:
        rng = np.random.default_rng(5)                   # (every rank draws the same GLOBAL case)
        NU, NI, Bl, K = 3001, 4003, 2048, 8              # table sizes not divisible by the world: ragged shards
        B = Bl * world
        U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
        b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
        uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
        nid = rng.integers(0, NI, (K, B)).astype(np.int32)
        uid[:, ::97] = 3; pid[:, ::53] = 7               # hot rows spread over every rank's slice
        lr = 0.002 if optk == "adam" else 0.05
        eng = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=rank, world=world, device=dev, slack=1.5, engine=engine, dedup=dedup)
        eng.force_collectives = True
        eng.U.write(U[rank::world]); eng.V.write(V[rank::world]); eng.b.write(b[rank::world])
        sl = slice(rank * Bl, (rank + 1) * Bl)
        tu, tp, tn = (torch.from_numpy(np.ascontiguousarray(x[:, sl])).to(dev) for x in (uid, pid, nid))
        torch.cuda.synchronize(); dist.barrier()
        eng.steps(tu[:6], tp[:6], tn[:6], plan_chunk=4, overlap=overlap)
        assert (eng._comm is not None) == (engine is None)
        for s in (6, 7):
            eng.step(tu[s], tp[s], tn[s])
        eng.check()
        oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
        tl = 0.0
        for s in range(K):
            step = orc.bpr_step if model == "bpr" else (lambda *a: orc.ucml_step(*a, do_censor=False))
            tl += float(step(U, V, b, uid[s], pid[s], nid[s], oo)[0])
        loss, _ = eng.loss_sums()
        tol = 5e-5 if optk == "adam" else 1e-5           # (tests/conftest.py: TOL_ADAM, TOL)
        assert abs(loss - tl) <= 1e-5 * abs(tl), (rank, model, optk, loss, tl)
        for got, want, nm in ((eng.U.read(), U[rank::world], "U"), (eng.V.read(), V[rank::world], "V"), (eng.b.read(), b[rank::world], "b")):
            assert rel_err(got, want) < tol, (rank, model, optk, nm, rel_err(got, want))
        dist.barrier()
        say(f"rccl-world{world} pairwise {model} {optk} D={D} engine={engine or 'library'} overlap={overlap} dedup={dedup}: ok")

    # ---- hot-item replication inside the library's engine (orx_sharded_pairwise_steps_hot): Zipf(1.05) item ids, the H most popular
    # items replicated on every rank (load_hot: an all-reduce of the owners' rows), ONE ncclAllReduce of the hot block per step, the
    # exchanged buckets sized for the cold share; replicas must stay identical on all ranks
    for model, optk, overlap in HOT_CASES:
        rng = np.random.default_rng(6)
        NU, NI, Bl, K, D, H = 3001, 4003, 2048, 6, 64, 256
        B = Bl * world
        U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
        b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
        w = 1.0 / np.arange(1, NI + 1) ** 1.05
        cdf = np.cumsum(w / w.sum())
        draw = lambda: np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1).astype(np.int32)
        uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid, nid = draw(), draw()
        cold = float(max(((pid[k] >= H).mean() + (nid[k] >= H).mean()) / 2 for k in range(K)))
        lr = 0.002 if optk == "adam" else 0.0005
        eng = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=rank, world=world, device=dev, slack=2.0, hot_items=H,
                                      hot_cold_fraction=min(1.0, cold * 1.5 + 0.05))
        eng.force_collectives = True
        eng.U.write(U[rank::world]); eng.V.write(V[rank::world]); eng.b.write(b[rank::world])
        sl = slice(rank * Bl, (rank + 1) * Bl)
        tu, tp, tn = (torch.from_numpy(np.ascontiguousarray(x[:, sl])).to(dev) for x in (uid, pid, nid))
        torch.cuda.synchronize(); dist.barrier()
        eng.steps(tu, tp, tn, plan_chunk=4, overlap=overlap)
        assert eng._comm is not None and eng._fast_hot
        eng.check()
        vh = torch.from_numpy(eng.Vh.read()).to(dev)
        lo, hi = vh.clone(), vh.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged across the ranks"
        eng.sync_hot()
        oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
        Uo, Vo, bo = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
        tl = sum(float(orc.bpr_step(Uo, Vo, bo, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
        loss, _ = eng.loss_sums()
        tol = 5e-5 if optk == "adam" else 1e-5
        assert abs(loss - tl) <= 1e-5 * abs(tl), (rank, "hot", optk, loss, tl)
        for got, want, nm in ((eng.U.read(), Uo[rank::world], "U"), (eng.V.read(), Vo[rank::world], "V"), (eng.b.read(), bo[rank::world], "b")):
            assert rel_err(got[:len(want)], want) < tol, (rank, "hot", optk, nm, rel_err(got[:len(want)], want))
        dist.barrier()
        say(f"rccl-world{world} pairwise hot-items {model} {optk} overlap={overlap}: ok")

    # ---- hybrid-parallel DLRM: embedding rows through all-to-all, dense gradients through all-reduce
    CFG = dict(m_spa=16, ln_emb=[1000, 37, 5000, 3, 250], ln_bot=[64, 16], ln_top=[128, 64, 1], dense_dim=13)
    CFG = dict(m_spa=32, ln_emb=[1000, 37, 5000, 3, 250], ln_bot=[64, 32], ln_top=[128, 64, 1], dense_dim=13)     # (m_spa 32: the rows are read where they arrive)
    for optk, loss_func, engine in DLRM_CASES:
        kw = dict(reference_compat=False, loss_func=loss_func)
        ref = DLRMOracle(seed=5, **dict(CFG, **kw))
        e = ShardedDLRM(rank=rank, world=world, device=dev, opt=optk, lr=0.05, slack=2.0, seed=5, engine=engine, **CFG, **kw)
        e.force_collectives = True
        e.load_embeddings(np.concatenate(ref.emb))
        for name, layers in (("bot", ref.bot), ("top", ref.top)):
            for l, (W, bb) in enumerate(layers):
                e.be.dense_param(name + "_w", l).write(W); e.be.dense_param(name + "_b", l).write(bb.reshape(1, -1))
        rng = np.random.default_rng(3)
        oo = {"sgd": lambda: orc.SGD(0.05), "adagrad": lambda: orc.Adagrad(0.05, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
        total = 0.0
        Bl = 256
        for s in range(3):
            dense = rng.normal(size=(Bl * world, 13)).astype(np.float32)
            sparse = np.stack([rng.integers(0, n, Bl * world) for n in CFG["ln_emb"]], 1).astype(np.int32)
            label = (rng.random(Bl * world) < 0.3).astype(np.float32)
            sl = slice(rank * Bl, (rank + 1) * Bl)
            torch.cuda.synchronize()
            e.steps(torch.from_numpy(dense[sl].copy()).to(dev)[None], torch.from_numpy(sparse[sl].copy()).to(dev)[None], torch.from_numpy(label[sl].copy()).to(dev)[None])
            assert (e._comm is not None) == (engine is None)
            total += float(ref.step(dense, sparse, label, oo))
        e.check()
        tol = 5e-5 if optk == "adam" else 2e-5           # (tests/test_sharded_dlrm_cpu.py)
        assert rel_err(e.local_embeddings(), np.concatenate(ref.emb)[rank::world]) < tol, (rank, optk)
        for name, layers in (("bot", ref.bot), ("top", ref.top)):
            for l, (W, bb) in enumerate(layers):
                assert rel_err(e.be.dense_param(name + "_w", l).read(), W) < tol, (rank, optk, name, l)
        assert abs(e.loss_sum() - total) < 1e-5 * abs(total), (rank, optk)
        dist.barrier()
        say(f"rccl-world{world} dlrm {optk} {loss_func} engine={engine or 'library'}: ok")

    dist.barrier()
    dist.destroy_process_group()
    say("RCCL_MULTIRANK_OK")


if __name__ == "__main__":
    main()
