#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_compose.py tests/test_gpu_reference_examples.py tests/test_gpu_api.py -m gpu -q --timeout 600 > gpurun_out/r3i_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3i_pytest.log
tail -n 40 gpurun_out/r3i_pytest.log
