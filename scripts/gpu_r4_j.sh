#!/bin/bash
mkdir -p gpurun_out/r4j
timeout 900 python -m pytest tests/test_gpu_sharded_dlrm.py tests/test_gpu_rccl_rank1.py tests/test_gpu_dlrm.py -q -x > gpurun_out/r4j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4j/pytest.log
tail -25 gpurun_out/r4j/pytest.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tests/rccl_multirank_worker.py > gpurun_out/r4j/multirank_w1.log 2>&1; tail -6 gpurun_out/r4j/multirank_w1.log
ORX_SHARD_RCCL_SELF=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 1 --sharded --model dlrm --fp16-mlp --steps 40 --warmup 10 2>/dev/null | tail -1 | cut -c1-700
timeout 300 python bench.py --gpus 1 --sharded --model dlrm --fp16-mlp --steps 40 --warmup 10 2>/dev/null | tail -1 | cut -c1-700
ORX_SHARD_ENGINE=python timeout 300 python bench.py --gpus 1 --sharded --model dlrm --fp16-mlp --steps 40 --warmup 10 2>/dev/null | tail -1 | cut -c1-400
