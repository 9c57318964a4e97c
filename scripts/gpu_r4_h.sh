#!/bin/bash
mkdir -p gpurun_out/r4h
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_sharded.py tests/test_gpu_compose.py tests/test_gpu_shard_engine.py tests/test_gpu_rccl_rank1.py -q -x > gpurun_out/r4h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4h/pytest.log
tail -15 gpurun_out/r4h/pytest.log
ORX_SHARD_RCCL_SELF=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --sharded --steps 20 --warmup 5 > gpurun_out/r4h/sharded_w1_rccl.json 2> gpurun_out/r4h/sharded_w1_rccl.err
tail -c 2500 gpurun_out/r4h/sharded_w1_rccl.json; tail -3 gpurun_out/r4h/sharded_w1_rccl.err
timeout 300 python bench.py --gpus 1 --sharded --steps 20 --warmup 5 > gpurun_out/r4h/sharded_w1_local.json 2> gpurun_out/r4h/sharded_w1_local.err
tail -c 1500 gpurun_out/r4h/sharded_w1_local.json
