#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_stress.py tests/test_gpu_fullsize.py tests/test_gpu_stepqueue.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_pointwise.py -m gpu -q -x --timeout 600 > gpurun_out/r3d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3d_pytest.log
for k in 20 200; do
  w=$((k/4)); [ $k = 20 ] && w=5
  timeout 300 python bench.py --steps $k --warmup $w --no-cpu-baseline --no-secondary > gpurun_out/r3d_bench_k$k.json 2> gpurun_out/r3d_bench_k$k.err
  ORX_PLAN_NO_PIPE=1 timeout 300 python bench.py --steps $k --warmup $w --no-cpu-baseline --no-secondary > gpurun_out/r3d_bench_k${k}_nopipe.json 2>> gpurun_out/r3d_bench_k$k.err
done
timeout 300 python bench.py --steps 200 --warmup 20 --zipf 1.05 --no-cpu-baseline --no-secondary > gpurun_out/r3d_bench_zipf.json 2> gpurun_out/r3d_bench_zipf.err
timeout 300 python bench.py --steps 200 --warmup 20 --model ucml --dim 128 --censor --no-cpu-baseline --no-secondary > gpurun_out/r3d_bench_c3.json 2> gpurun_out/r3d_bench_c3.err
tail -n 6 gpurun_out/r3d_pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d_bench_*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline',{})
        print(f, 'ms/step', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r.get('kernel_us',0),2), 'other', {k:round(v,1) for k,v in r.get('other_kernels_us',{}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
