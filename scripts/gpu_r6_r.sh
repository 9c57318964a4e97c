#!/bin/bash
# round 6: the suites that drive the pointwise step through other entry points, after the reducer workgroups went into GMF's launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6r; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_api.py tests/test_gpu_compose.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py tests/test_gpu_stepqueue.py tests/test_gpu_modules.py tests/test_gpu_reference_examples.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
