#!/bin/bash
cd /root/repo
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', j['ms_per_step'], {k:(v['ms_per_step'], v['roofline'].get('kernel_us')) for k,v in j['secondary'].items()})"; done
timeout 600 python bench.py --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*'
timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_rows_sorted.py -x -q 2>&1 | tail -n 2
