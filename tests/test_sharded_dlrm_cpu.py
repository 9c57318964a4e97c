"""The hybrid-parallel DLRM step (openrec_amd/sharded_dlrm.py) on CPU: world size 2 over gloo (and world
size 1 in-process) must reproduce the single-process oracle on the global batch.  Compute is the oracle here
(the HIP building blocks are parity-tested on the GPU); under test: the id routing, the bucket plan, the
all-to-all / all-reduce sequence and the global-batch loss scaling."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_err
from oracle.dlrm_oracle import DLRMOracle

CFG = dict(m_spa=8, ln_emb=[13, 7, 40, 5], ln_bot=[16, 8], ln_top=[32, 16, 1], dense_dim=6)


def _case(steps=3, B=48, seed=3):
    rng = np.random.default_rng(seed)
    out = []
    for s in range(steps):
        dense = rng.normal(size=(B, CFG["dense_dim"])).astype(np.float32)
        sparse = np.stack([rng.integers(0, n, B) for n in CFG["ln_emb"]], 1).astype(np.int32)
        sparse[:7, 2] = 11                                    # duplicates inside a step
        label = (rng.random(B) < 0.3).astype(np.float32)
        out.append((dense, sparse, label))
    return out


def _run_rank(rank, world, port, optk, compat, out):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_dlrm_ref_backend import OracleDLRMBackend
    from openrec_amd.sharded_dlrm import ShardedDLRM
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(CFG, reference_compat=compat, loss_func="bce")
    ref = DLRMOracle(seed=5, **cfg)
    be = OracleDLRMBackend(cfg, optk, 0.05, seed=5)
    eng = ShardedDLRM(rank=rank, world=world, device=torch.device("cpu"), backend=be, opt=optk, lr=0.05, slack=2.0, **CFG)
    eng.load_embeddings(np.concatenate(ref.emb))
    for dense, sparse, label in _case():
        per = label.shape[0] // world
        sl = slice(rank * per, (rank + 1) * per)
        eng.step(torch.from_numpy(dense[sl].copy()), torch.from_numpy(sparse[sl].copy()), torch.from_numpy(label[sl].copy()))
    eng.check()
    np.savez(out % rank, emb=eng.local_embeddings(), loss=eng.loss_sum(),
             **{"p%d" % k: p for k, (_, p) in enumerate(be._params())})
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("optk,compat", [("sgd", False), ("adagrad", False), ("sgd", True), ("adam", False)])
def test_sharded_dlrm_equals_single_process(tmp_path, world, optk, compat):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from sharded_dlrm_ref_backend import make_opt
    out = str(tmp_path / "r%d.npz")
    if world == 1:
        _run_rank(0, 1, 0, optk, compat, out)
    else:
        mp.spawn(_run_rank, args=(world, _free_port(), optk, compat, out), nprocs=world, join=True)
    ref = DLRMOracle(seed=5, **dict(CFG, reference_compat=compat, loss_func="bce"))
    opt = make_opt(optk, 0.05)
    total = 0.0
    for dense, sparse, label in _case():
        total += float(ref.step(dense, sparse, label, opt))
    comb = np.concatenate(ref.emb)
    dense_params = [p for layers in (ref.bot, ref.top) for Wb in layers for p in Wb]
    for r in range(world):
        g = np.load(out % r)
        assert rel_err(g["emb"], comb[r::world]) < 2e-5
        for k, p in enumerate(dense_params):
            assert rel_err(g["p%d" % k], p) < 2e-5, k
        assert abs(float(g["loss"]) - total) < 1e-5 * abs(total)
