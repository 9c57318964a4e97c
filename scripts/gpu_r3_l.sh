#!/bin/bash
# round 3, call l: the sharded engine inside the library (orx_sharded_pairwise_steps)
cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_shard_engine.py tests/test_gpu_rccl_rank1.py tests/test_gpu_sharded.py tests/test_gpu_c4_shapes.py -x -q > gpurun_out/r3l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3l_pytest.log
tail -n 25 gpurun_out/r3l_pytest.log
echo "--- world-1 engine, no collectives"
timeout 600 python bench.py --sharded --steps 100 --warmup 20 --no-secondary 2>&1 | tail -n 1 | tee gpurun_out/r3l_bench_sharded_local.json | grep -o '"ms_per_step": [0-9.]*'
echo "--- world-1 through RCCL (loop-back)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --sharded --steps 100 --warmup 20 --no-secondary 2>&1 | tail -n 1 | tee gpurun_out/r3l_bench_sharded_rccl1.json | grep -o '"ms_per_step": [0-9.]*'
