"""The sharded engines on W REAL ranks: torch.distributed.run --nproc-per-node W with backend "nccl" (RCCL over xGMI), one
rank per GPU, the way the driver launches `bench.py --gpus N` -- see tests/rccl_multirank_worker.py.  Skipped where the box has
fewer than W GPUs (the test tiers' boxes have one: there the same exchange plan is covered by tests/test_gpu_rccl_rank1.py, the
in-process virtual clusters of tests/test_gpu_shard_engine.py and the gloo world-2/4 tests on CPU)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gpus():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_engines_on_rccl_with_w_ranks(world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs on one node, this box has {_gpus()}")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rccl_multirank_worker import PAIRWISE_CASES, HOT_CASES, DLRM_CASES
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_multirank_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "RCCL_MULTIRANK_OK" in r.stdout, r.stdout[-3000:] + "\n---- stderr ----\n" + r.stderr[-3000:]
    assert r.stdout.count(": ok") == len(PAIRWISE_CASES) + len(HOT_CASES) + len(DLRM_CASES), r.stdout
