#!/bin/bash
# usage: scripts/plan_probe.sh <tag> [env assignments...] -- per-kernel summary of bench K=200 and K=20 under rocprofv3
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for K in 200 20; do
  W=$((K/10)); [ $K -eq 20 ] && W=5
  env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/pp_${TAG}_$K -o t -- python $REPO/bench.py --steps $K --warmup $W --no-cpu-baseline > $OUT/pp_${TAG}_$K.log 2>&1
  echo "== $TAG K=$K $@"; python $REPO/scripts/trace_summary.py $OUT/pp_${TAG}_$K/t_kernel_trace.csv 8 | grep -v "at::native\|rocclr\|init_uniform"
  grep -o '"ms_per_step": [0-9.]*' $OUT/pp_${TAG}_$K.log
done
