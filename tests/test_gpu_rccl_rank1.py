"""The sharded engines on the REAL collective backend: one rank launched by torch.distributed.run with backend "nccl"
(RCCL on ROCm), the way the driver launches `bench.py --gpus N` -- see tests/rccl_rank1_worker.py.  (The multi-rank
exchange plan itself is covered by the gloo world-2 tests on CPU and the in-process virtual clusters on the GPU; an 8-GPU
node is not available to the test tiers.)"""
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


@pytest.mark.parametrize("rccl_self", ["1", "0"])
def test_sharded_engines_on_rccl_with_one_rank(rccl_self):
    # ORX_SHARD_RCCL_SELF=1: the rank's own block goes through ncclSend / ncclRecv too (otherwise a device copy): with one rank
    # that is what puts the library's RCCL groups on the wire at all
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", ORX_SHARD_RCCL_SELF=rccl_self)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "rccl_rank1_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_RANK1_OK" in r.stdout, r.stdout[-3000:] + "\n---- stderr ----\n" + r.stderr[-3000:]
    assert r.stdout.count(": ok") == 16, r.stdout
