#!/bin/bash
# round 4, call A: GPU suite on the pairing build, then the headline with / without pairing at K = 20 and K = 200
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -x > gpurun_out/r4a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.log
tail -15 gpurun_out/r4a/pytest.log
for k in 20 200; do
  w=5; [ $k = 200 ] && w=20
  ORX_PLAN_DEBUG=1 timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4a/pair_k$k.json 2> gpurun_out/r4a/pair_k$k.err
  ORX_NO_PAIR=1 timeout 300 python bench.py --steps $k --warmup $w --no-secondary --no-cpu-baseline > gpurun_out/r4a/nopair_k$k.json 2> gpurun_out/r4a/nopair_k$k.err
done
for f in gpurun_out/r4a/*.json; do echo $f; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print(' ms/step %.5f  value %.4g  fused_us %.2f frac %.3f other %s'%(d['ms_per_step'],d['value'],r['kernel_us'],r['frac'],{k:round(v,1) for k,v in r['other_kernels_us'].items()}))"; done
grep -h "orx plan" gpurun_out/r4a/pair_k20.err | head -3
