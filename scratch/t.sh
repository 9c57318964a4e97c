cd /root/repo
timeout 900 python -m pytest tests/test_gpu_stress.py tests/test_gpu_pairwise.py -x -q 2>&1 | tail -3
python bench.py --model ucml --dim 128 --censor --opt adam --steps 64 --warmup 64 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ucml128 censor adam ms_per_step', d['ms_per_step'])"
ORX_ADAM_DENSE=1 python bench.py --model ucml --dim 128 --censor --opt adam --steps 16 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dense: ucml128 censor adam ms_per_step', d['ms_per_step'])"
