"""Worker of tests/test_gpu_rccl_rank1.py, launched as ONE rank by torch.distributed.run with backend "nccl" (= RCCL):
the row-sharded pairwise engine (K-step planned path and the per-step path) and the hybrid-parallel DLRM step, every
exchange forced through torch.distributed (`force_collectives`), against the NumPy oracles.  What this holds that the
in-process virtual cluster cannot: the library's private stream and RCCL's collectives order correctly through the
engine's torch stream, on the real backend the 8-GPU bench uses."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    import torch
    import torch.distributed as dist
    from openrec_amd import sharded
    from openrec_amd.sharded_dlrm import ShardedDLRM
    from oracle import numpy_oracle as orc
    from oracle.dlrm_oracle import DLRMOracle

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 1
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=dev)
    assert dist.get_backend() == "nccl"

    # ---- row-sharded BPR / UCML: 6 steps through the planned K-step path, 2 through the per-step path
    # engine None: orx_sharded_pairwise_steps (the whole K-step loop and the ncclSend / ncclRecv groups inside the library, on an
    # orx_comm made from an id of its own), overlapped halves and sequential; "python": the per-phase path over torch.distributed
    for model, optk, D, engine, overlap in (("bpr", "sgd", 64, None, None), ("bpr", "sgd", 64, None, False), ("bpr", "sgd", 64, "python", None),
                                            ("bpr", "adagrad", 64, None, None), ("bpr", "adagrad", 64, "python", None),
                                            ("ucml", "sgd", 128, None, None), ("bpr", "adam", 64, None, None), ("bpr", "adam", 64, None, False),
                                            ("bpr", "adam", 64, "python", None)):
        rng = np.random.default_rng(5)
        NU, NI, B, K = 3000, 4000, 4096, 8
        U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
        b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
        uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid = rng.integers(0, NI, (K, B)).astype(np.int32)
        nid = rng.integers(0, NI, (K, B)).astype(np.int32)
        uid[:, :9] = 3
        lr = 0.002 if optk == "adam" else 0.05
        eng = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=1.0, engine=engine,
                                      dedup=True if overlap is False else None)      # (the sequential library runs also force the dedup plan)
        eng.force_collectives = True
        eng.U.write(U); eng.V.write(V); eng.b.write(b)
        tu, tp, tn = (torch.from_numpy(x).to(dev) for x in (uid, pid, nid))
        torch.cuda.synchronize()
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
        tol = 5e-5 if optk == "adam" else 1e-5          # (tests/conftest.py: TOL_ADAM, TOL)
        assert abs(loss - tl) <= 1e-5 * abs(tl), (model, optk, loss, tl)
        for got, want, nm in ((eng.U.read(), U, "U"), (eng.V.read(), V, "V"), (eng.b.read(), b, "b")):
            assert rel_err(got, want) < tol, (model, optk, nm, rel_err(got, want))
        print(f"rccl-rank1 pairwise {model} {optk} D={D} engine={engine or 'library'} overlap={overlap}: ok", flush=True)

    # ---- hot-item replication inside the library's engine (orx_sharded_pairwise_steps_hot): Zipf item ids, the replica filled by
    # load_hot() (an all-reduce over the process group), the per-step all-reduce of the hot block through ncclAllReduce
    for model, optk, overlap in (("bpr", "sgd", None), ("bpr", "adagrad", False), ("bpr", "adam", None)):
        rng = np.random.default_rng(6)
        NU, NI, B, K, D, H = 3000, 4000, 4096, 6, 64, 256
        U = rng.uniform(-.05, .05, (NU, D)).astype(np.float32); V = rng.uniform(-.05, .05, (NI, D)).astype(np.float32)
        b = rng.uniform(-.05, .05, (NI, 1)).astype(np.float32)
        w = 1.0 / np.arange(1, NI + 1) ** 1.05
        cdf = np.cumsum(w / w.sum())
        draw = lambda: np.minimum(np.searchsorted(cdf, rng.random((K, B))), NI - 1).astype(np.int32)
        uid = rng.integers(0, NU, (K, B)).astype(np.int32); pid, nid = draw(), draw()
        lr = 0.002 if optk == "adam" else 0.0005
        eng = sharded.ShardedPairwise(model, optk, NU, NI, D, lr=lr, rank=0, world=1, device=dev, slack=3.0, hot_items=H)
        eng.force_collectives = True
        eng.U.write(U); eng.V.write(V); eng.b.write(b)
        tu, tp, tn = (torch.from_numpy(x).to(dev) for x in (uid, pid, nid))
        torch.cuda.synchronize()
        eng.steps(tu, tp, tn, plan_chunk=4, overlap=overlap)
        assert eng._comm is not None and eng._fast_hot
        eng.check()
        eng.sync_hot()
        oo = {"sgd": lambda: orc.SGD(lr), "adagrad": lambda: orc.Adagrad(lr, 0.1, 1e-7), "adam": lambda: orc.AdamTFSparse(lr)}[optk]()
        Uo, Vo, bo = U.astype(np.float64), V.astype(np.float64), b.astype(np.float64)
        tl = sum(float(orc.bpr_step(Uo, Vo, bo, uid[s], pid[s], nid[s], oo)[0]) for s in range(K))
        loss, _ = eng.loss_sums()
        tol = 5e-5 if optk == "adam" else 1e-5
        assert abs(loss - tl) <= 1e-5 * abs(tl), (model, optk, loss, tl)
        for got, want, nm in ((eng.U.read(), Uo, "U"), (eng.V.read(), Vo, "V"), (eng.b.read(), bo, "b")):
            assert rel_err(got, want) < tol, ("hot", model, optk, nm, rel_err(got, want))
        print(f"rccl-rank1 pairwise hot-items {model} {optk} overlap={overlap}: ok", flush=True)

    # ---- hybrid-parallel DLRM: embedding rows through all-to-all, dense gradients through all-reduce
    CFG = dict(m_spa=16, ln_emb=[1000, 37, 5000, 3, 250], ln_bot=[64, 16], ln_top=[128, 64, 1], dense_dim=13)
    for optk, loss_func, engine in (("sgd", "bce", "python"), ("adam", "mse", "python"), ("sgd", "bce", None), ("adam", "mse", None)):
        if engine is None:      # the library's engine (orx_sharded_dlrm_steps) reads the rows where they arrive: m_spa >= 32
            CFG = dict(CFG, m_spa=32, ln_bot=[64, 32])
        kw = dict(reference_compat=False, loss_func=loss_func)
        ref = DLRMOracle(seed=5, **dict(CFG, **kw))
        e = ShardedDLRM(rank=0, world=1, device=dev, opt=optk, lr=0.05, slack=2.0, seed=5, engine=engine, **CFG, **kw)
        e.force_collectives = True
        e.load_embeddings(np.concatenate(ref.emb))
        for name, layers in (("bot", ref.bot), ("top", ref.top)):
            for l, (W, bb) in enumerate(layers):
                e.be.dense_param(name + "_w", l).write(W); e.be.dense_param(name + "_b", l).write(bb.reshape(1, -1))
        rng = np.random.default_rng(3)
        oo = {"sgd": lambda: orc.SGD(0.05), "adam": lambda: orc.AdamTFSparse(0.05)}[optk]()
        total = 0.0
        for s in range(3):
            dense = rng.normal(size=(512, 13)).astype(np.float32)
            sparse = np.stack([rng.integers(0, n, 512) for n in CFG["ln_emb"]], 1).astype(np.int32)
            label = (rng.random(512) < 0.3).astype(np.float32)
            torch.cuda.synchronize()
            e.steps(torch.from_numpy(dense).to(dev)[None], torch.from_numpy(sparse).to(dev)[None], torch.from_numpy(label).to(dev)[None])
            assert (e._comm is not None) == (engine is None)
            total += float(ref.step(dense, sparse, label, oo))
        e.check()
        tol = 5e-5 if optk == "adam" else 1e-5
        assert rel_err(e.local_embeddings(), np.concatenate(ref.emb)) < tol, optk
        for name, layers in (("bot", ref.bot), ("top", ref.top)):
            for l, (W, bb) in enumerate(layers):
                assert rel_err(e.be.dense_param(name + "_w", l).read(), W) < tol, (optk, name, l)
        assert abs(float(e.loss_accum.item()) - total) < 1e-5 * abs(total)
        print(f"rccl-rank1 dlrm {optk} {loss_func} engine={engine or 'library'}: ok", flush=True)

    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_RANK1_OK", flush=True)


if __name__ == "__main__":
    main()
