cd /root/repo
timeout 900 python -m pytest tests/test_gpu_api.py tests/test_gpu_stress.py tests/test_gpu_pointwise.py tests/test_gpu_dlrm.py -x -q 2>&1 | tail -4
