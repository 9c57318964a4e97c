cd /root/repo
timeout 900 python -m pytest tests/test_gpu_pointwise.py -q 2>&1 | grep -E "^E   +assert|AssertionError|passed|failed|Error|^tests/.*:[0-9]+:" | head -20
