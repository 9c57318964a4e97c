#!/bin/bash
# full GPU suite on the pairing build + the secondary pairwise workloads with / without pairing
mkdir -p gpurun_out/r4f
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r4f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4f/pytest.log
tail -4 gpurun_out/r4f/pytest.log
run() { tag=$1; shift
  timeout 300 python bench.py --no-secondary --no-cpu-baseline "$@" > gpurun_out/r4f/$tag.json 2> gpurun_out/r4f/$tag.err
  ORX_NO_PAIR=1 timeout 300 python bench.py --no-secondary --no-cpu-baseline "$@" > gpurun_out/r4f/${tag}_nopair.json 2> gpurun_out/r4f/${tag}_nopair.err
}
run c2_k20 --steps 20 --warmup 5
run c3 --model ucml --dim 128 --censor --steps 200 --warmup 20
run adagrad --opt adagrad --steps 200 --warmup 20
run zipf --zipf 1.05 --steps 200 --warmup 20
run small --users 100000 --items 100000 --steps 200 --warmup 20
run d32 --dim 32 --steps 200 --warmup 20
for f in gpurun_out/r4f/*.json; do python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
    print('%-28s ms/step %.5f fused_us %.2f frac %.3f other %s'%('$f'.split('/')[-1],d['ms_per_step'],r['kernel_us'],r['frac'],{k:round(v,1) for k,v in r['other_kernels_us'].items()}))
except Exception as e: print('$f', 'ERR', e)"; done
python scripts/k20_timeline.py gpurun_out/q_r4e_k20/q_kernel_trace.csv 2>/dev/null | tail -3
