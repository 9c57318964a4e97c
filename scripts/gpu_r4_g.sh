#!/bin/bash
mkdir -p gpurun_out/r4g
timeout 900 python -m pytest tests/test_gpu_compose.py tests/test_gpu_pointwise.py tests/test_gpu_reference_examples.py tests/test_gpu_dlrm.py -q -x > gpurun_out/r4g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4g/pytest.log
tail -25 gpurun_out/r4g/pytest.log
run() { tag=$1; shift
  timeout 300 python bench.py --no-secondary --no-cpu-baseline "$@" > gpurun_out/r4g/$tag.json 2> gpurun_out/r4g/$tag.err
}
run c2_k20 --steps 20 --warmup 5
run c2_k200 --steps 200 --warmup 20
run adagrad --opt adagrad --steps 200 --warmup 20
run zipf --zipf 1.05 --steps 200 --warmup 20
run small --users 100000 --items 100000 --steps 200 --warmup 20
for f in gpurun_out/r4g/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-28s ms/step %.5f fused_us %.2f frac %.3f other %s'%('$f'.split('/')[-1],d['ms_per_step'],r['kernel_us'],r['frac'],{k:round(v,1) for k,v in r['other_kernels_us'].items()}))
except Exception as e: print('$f', 'ERR', e)"; done
