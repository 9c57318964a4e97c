cd /root/repo
timeout 600 python -m pytest tests/test_gpu_stress.py -x -q -k "skewed" 2>&1 | grep -E "^E   +assert|AssertionError|passed|failed|Error" | head
python bench.py --opt adam --zipf 1.05 --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bpr adam zipf ms_per_step', d['ms_per_step'])"
