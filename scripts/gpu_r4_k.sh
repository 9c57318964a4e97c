#!/bin/bash
mkdir -p gpurun_out/r4k
timeout 1200 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_shard_engine.py tests/test_gpu_stepqueue.py tests/test_gpu_stress.py tests/test_metrics.py tests/test_gpu_sampler.py tests/test_gpu_rows_sorted.py tests/test_gpu_reference_examples.py tests/test_gpu_rccl_rank1.py tests/test_gpu_rccl_multirank.py -q -x > gpurun_out/r4k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4k/pytest.log
tail -12 gpurun_out/r4k/pytest.log
