#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for k in 20 40 64; do
  timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r3e_bench_k$k.json 2> gpurun_out/r3e_bench_k$k.err
  ORX_PLAN_NO_PIPE=1 timeout 300 python bench.py --steps $k --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r3e_bench_k${k}_nopipe.json 2>> gpurun_out/r3e_bench_k$k.err
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > gpurun_out/r3e_bench_k20_b.json 2>> gpurun_out/r3e_bench_k20.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e_bench_*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline',{})
        print(f, 'us/step', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r.get('kernel_us',0),2), 'other', {k:round(v,1) for k,v in r.get('other_kernels_us',{}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
