#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_stress.py tests/test_gpu_compose.py tests/test_gpu_stepqueue.py tests/test_gpu_sampler.py -m gpu -q -x --timeout 600 > gpurun_out/r3h_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3h_pytest.log
for m in wrmf gmf; do
  timeout 300 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3h_$m.json 2>> gpurun_out/r3h.err
  ORX_FORCE_FALLBACK=2 timeout 300 python bench.py --model $m --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r3h_${m}_noinline.json 2>> gpurun_out/r3h.err
  timeout 300 python bench.py --model $m --steps 200 --warmup 20 --zipf 1.05 --no-cpu-baseline > gpurun_out/r3h_${m}_zipf.json 2>> gpurun_out/r3h.err
done
timeout 300 python bench.py --model wrmf --opt adam --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/r3h_wrmf_adam.json 2>> gpurun_out/r3h.err
tail -n 4 gpurun_out/r3h_pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3h_*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline',{})
        print(f, 'us/step', round(d['ms_per_step']*1e3,2), 'kernel_us', round(r.get('kernel_us',0),2), 'other', {k:round(v,1) for k,v in r.get('other_kernels_us',{}).items()})
    except Exception as e:
        print(f, 'ERR', e)
PY
