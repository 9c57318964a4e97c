#!/bin/bash
mkdir -p gpurun_out/r4d
timeout 600 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_fuzz.py tests/test_gpu_stepqueue.py tests/test_gpu_api.py -q -x > gpurun_out/r4d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4d/pytest.log
tail -5 gpurun_out/r4d/pytest.log
for k in 20 200; do
  w=5; [ $k = 200 ] && w=20
  ORX_PLAN_DEBUG=1 timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4d/pair_k$k.json 2> gpurun_out/r4d/pair_k$k.err
  ORX_NO_PAIR=1 timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4d/nopair_k$k.json 2> gpurun_out/r4d/nopair_k$k.err
done
for f in gpurun_out/r4d/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print(' ms/step %.5f  value %.4g  fused_us %.2f frac %.3f other %s'%(d['ms_per_step'],d['value'],r['kernel_us'],r['frac'],{k:round(v,1) for k,v in r['other_kernels_us'].items()}))"; done
grep -h "orx plan" gpurun_out/r4d/pair_k20.err | tail -1
for k in 20 200; do
ORX_PLAN_TIMING=1 timeout 300 python bench.py --steps $k --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "plan timing" | tail -1
done
bash scripts/quick_stats.sh r4d_k20 --steps 20 --warmup 5 --no-secondary 2>&1 | tail -16
