cd /root/repo
ex() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], 'ms_per_step', round(d['ms_per_step'],3), 'samples/s', int(d['value']))" "$1"; }
timeout 600 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_sharded_dlrm.py -x -q 2>&1 | tail -3
timeout 200 python bench.py --model dlrm --opt adam --fp16-mlp --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | ex lazy_adam
