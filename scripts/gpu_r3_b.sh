#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export ORX_TEST_RECORD=$PWD/gpurun_out/r3b_records.jsonl
rm -f "$ORX_TEST_RECORD"
timeout 1500 python -m pytest tests/test_gpu_rows_sorted.py tests/test_gpu_dlrm.py tests/test_gpu_sharded_dlrm.py -m gpu -q --timeout 600 > gpurun_out/r3b_pytest1.log 2>&1
echo "pytest1 rc=$?" >> gpurun_out/r3b_pytest1.log
timeout 900 python -m pytest tests/test_gpu_c5_shapes.py -m gpu -q --timeout 800 -k fp16 > gpurun_out/r3b_pytest_c5.log 2>&1
echo "pytest_c5 rc=$?" >> gpurun_out/r3b_pytest_c5.log
export ORX_TEST_RECORD=$PWD/gpurun_out/r3b_records_atomics.jsonl
rm -f "$ORX_TEST_RECORD"
ORX_ROWS_ATOMICS=1 timeout 600 python -m pytest tests/test_gpu_dlrm.py -m gpu -q --timeout 600 -k "long_gaps" > gpurun_out/r3b_pytest_atomics.log 2>&1
tail -n 8 gpurun_out/r3b_pytest1.log gpurun_out/r3b_pytest_c5.log gpurun_out/r3b_pytest_atomics.log
