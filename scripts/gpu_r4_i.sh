#!/bin/bash
mkdir -p gpurun_out/r4i
for sh in default 14 13; do
  for k in 20 200; do
    w=5; [ $k = 200 ] && w=20
    if [ $sh = default ]; then unset ORX_PLAN_SHIFT; else export ORX_PLAN_SHIFT=$sh; fi
    timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4i/s${sh}_k$k.json 2>/dev/null
    ORX_PLAN_TIMING=1 timeout 300 python bench.py --steps $k --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "plan timing" | tail -1
  done
done
unset ORX_PLAN_SHIFT
for f in gpurun_out/r4i/*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('%-22s ms/step %.5f fused_us %.2f other %s'%('$f'.split('/')[-1],d['ms_per_step'],r['kernel_us'],{k:round(v,1) for k,v in r['other_kernels_us'].items()}))"; done
timeout 300 python -m pytest tests/test_gpu_stress.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -2
