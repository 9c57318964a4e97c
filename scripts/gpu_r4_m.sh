#!/bin/bash
mkdir -p gpurun_out/r4m
for rep in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r4m/base_$rep.json 2>/dev/null
  HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > gpurun_out/r4m/poll_$rep.json 2>/dev/null
done
HSA_ENABLE_INTERRUPT=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > gpurun_out/r4m/poll_k200.json 2>/dev/null
for f in gpurun_out/r4m/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s ms/step %.5f fused_us %.2f'%('$f'.split('/')[-1],d['ms_per_step'],r['kernel_us']))"; done
