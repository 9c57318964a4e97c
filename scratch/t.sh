cd /root/repo
timeout 600 python -m pytest tests/test_gpu_dlrm.py -x -q 2>&1 | tail -2
python bench.py --model dlrm --fp16-mlp --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dlrm sgd fp16 ms_per_step', round(d['ms_per_step'],4), 'gemm_ms', round(d['roofline']['gemm_ms_per_step'],4), 'TF/s', round(d['roofline']['achieved'],1))"
