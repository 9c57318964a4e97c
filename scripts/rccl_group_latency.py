#!/usr/bin/env python3
"""What one ncclSend / ncclRecv group of the row-sharded engine costs on ONE rank (its own block through RCCL, ORX_SHARD_RCCL_SELF=1):
the engine's own exchange (orx_comm_ping) at sizes from 4 KB to 32 MiB -- the fixed cost of a group (host call + RCCL kernel launch) is
the small-size end, the slope is what RCCL's self-send kernel moves per second.  Run under torch.distributed.run with one rank."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ORX_SHARD_RCCL_SELF", "1")
import torch
import torch.distributed as dist


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    from openrec_amd import sharded
    eng = sharded.ShardedPairwise("bpr", "sgd", 100000, 100000, 64, lr=0.05, rank=0, world=1, device=torch.device("cuda:0"), seed=0)
    eng.force_collectives = True
    rows = []
    eng.comm_ping(1 << 20, 3)
    for nbytes in (4096, 65536, 1 << 20, 4 << 20, 16 << 20, 32 << 20):
        t0 = time.perf_counter()
        p = eng.comm_ping(nbytes, 50)
        host = (time.perf_counter() - t0) / 51 * 1e6
        rows.append(dict(bytes=nbytes, us_per_group_device=round(p["us_per_all_to_all"], 2), us_per_call_host_incl_sync=round(host, 2),
                         GBps=round(p["GBps_out"], 1)))
        print(rows[-1], flush=True)
    print(json.dumps(rows))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
