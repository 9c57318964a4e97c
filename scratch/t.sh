cd /root/repo
timeout 900 python -m pytest tests/test_gpu_api.py -x -q 2>&1 | tail -5
