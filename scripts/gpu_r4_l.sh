#!/bin/bash
mkdir -p gpurun_out/r4l
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py tests/test_gpu_stepqueue.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
for rep in 1 2; do
for k in 20 200; do
  w=5; [ $k = 200 ] && w=20
  timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4l/side_k${k}_$rep.json 2>/dev/null
  ORX_PLAN_NO_SIDE=1 timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4l/noside_k${k}_$rep.json 2>/dev/null
done; done
for f in gpurun_out/r4l/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s ms/step %.5f fused_us %.2f other %s'%('$f'.split('/')[-1],d['ms_per_step'],r['kernel_us'],{k:round(v,1) for k,v in r['other_kernels_us'].items()}))"; done
