#!/bin/bash
# round 3, call s: the hybrid-parallel DLRM step with the exchanged rows read in place
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sharded_dlrm.py tests/test_gpu_dlrm.py tests/test_gpu_rccl_rank1.py -x -q 2>&1 | tail -n 6
echo "--- world 1, no collectives"
timeout 600 python bench.py --sharded --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | tee gpurun_out/r3s_dlrm_sharded_local.json | grep -o '"ms_per_step": [0-9.]*'
echo "--- world 1 through RCCL"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --sharded --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline 2>&1 | tail -n 1 | tee gpurun_out/r3s_dlrm_sharded_rccl1.json | grep -o '"ms_per_step": [0-9.]*'
