#!/bin/bash
# references per thread of plan_part_kernel (chunks of 256 x R references: fewer chunks = fewer same-address atomics on the bucket
# counters and cursors), one box: R = 8 (the build), 16, 32
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), d['roofline'].get('other_kernels_us'))"; }
for i in 1 2; do
  echo -n "R=8  K=20: "; one --steps 20 --warmup 5
  echo -n "R=16 K=20: "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_refs16.so one --steps 20 --warmup 5
  echo -n "R=32 K=20: "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_refs32.so one --steps 20 --warmup 5
done
echo -n "R=8  K=200: "; one --steps 200 --warmup 5
echo -n "R=16 K=200: "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_refs16.so one --steps 200 --warmup 5
echo -n "R=32 K=200: "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_refs32.so one --steps 200 --warmup 5
for L in refs16 refs32; do
ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_$L.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_k20c -o p -- python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find gpurun_out/r5_k20c -name '*kernel_trace.csv' | head -1)
echo "--- $L"; python scripts/k20_timeline.py "$f" | grep plan_
rm -rf gpurun_out/r5_k20c
done
