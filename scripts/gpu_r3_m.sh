#!/bin/bash
# round 3, call m: kernel times of the sharded engine at world 1 (local and RCCL loop-back)
cd /tmp && export TMPDIR=/tmp
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=/root/repo
mkdir -p $R/gpurun_out/r3m
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_local -- python $R/bench.py --sharded --steps 100 --warmup 20 --no-secondary > $R/gpurun_out/r3m/local.json 2> /tmp/err1.log
f=$(find /tmp/prof_local -name '*kernel_stats.csv' | head -n 1); cp "$f" $R/gpurun_out/r3m/local_kernel_stats.csv; head -n 14 "$f"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rccl -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 $R/bench.py --gpus 1 --sharded --steps 100 --warmup 20 --no-secondary > $R/gpurun_out/r3m/rccl1.json 2> /tmp/err2.log
for f in $(find /tmp/prof_rccl -name '*kernel_stats.csv'); do echo $f; head -n 16 "$f"; cp "$f" $R/gpurun_out/r3m/rccl1_kernel_stats.csv; done
tail -n 3 /tmp/err2.log
