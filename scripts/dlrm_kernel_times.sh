#!/bin/bash
# per-kernel average durations of the DLRM fp16 step under the given environment: scripts/dlrm_kernel_times.sh <tag> <kernel-name-substring>...
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 20 --warmup 5 > $OUT/${TAG}_prof.log 2>&1
cd $REPO
for k in "$@"; do grep "$k" $OUT/prof_$TAG/b_kernel_stats.csv | awk -F, -v t=$TAG '{printf "%s %-40s avg %.1f us\n", t, substr($1,1,40), $(NF-4)/1000}'; done
