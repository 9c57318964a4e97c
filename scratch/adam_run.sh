cd /root/repo
timeout 600 python -m pytest tests/test_gpu_stress.py -x -q -k "lazy_adam" 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_pairwise.py -x -q -k "golden" 2>&1 | tail -5
timeout 200 python bench.py --opt adam --steps 64 --warmup 64 2>&1 | tail -2
ORX_ADAM_DENSE=1 timeout 200 python bench.py --opt adam --steps 64 --warmup 64 2>&1 | tail -2
