#!/bin/bash
# documentation runs on the final build: the kernels of one K = 20 call in order; the multi-rank RCCL worker with one rank
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4_k20 -o q -- python bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-secondary > gpurun_out/r4_k20.log 2>&1
f=$(find gpurun_out/r4_k20 -name '*kernel_trace.csv' | head -1)
python scripts/k20_timeline.py "$f" > gpurun_out/r4_c2_k20_timeline.txt; cat gpurun_out/r4_c2_k20_timeline.txt
rm -rf gpurun_out/r4_k20
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 tests/rccl_multirank_worker.py > gpurun_out/r4_rccl_multirank_world1.log 2>&1; tail -5 gpurun_out/r4_rccl_multirank_world1.log
