#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), d['roofline'].get('other_kernels_us'))"; }
for i in 1 2 3; do
  echo -n "T=256 K=20: "; one --steps 20 --warmup 5
  echo -n "T=512 K=20: "; ORX_PLAN_RANGE_T=512 one --steps 20 --warmup 5
done
echo -n "T=256 K=200: "; one --steps 200 --warmup 5
echo -n "T=512 K=200: "; ORX_PLAN_RANGE_T=512 one --steps 200 --warmup 5
ORX_PLAN_RANGE_T=512 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_k20c -o p -- python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find gpurun_out/r5_k20c -name '*kernel_trace.csv' | head -1)
python scripts/k20_timeline.py "$f" | grep plan_
rm -rf gpurun_out/r5_k20c
ORX_PLAN_RANGE_T=512 timeout 300 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pairwise.py -x -q -m gpu 2>&1 | tail -3
