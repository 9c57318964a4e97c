#!/bin/bash
# usage: scripts/quick_stats.sh <tag> <bench args...>   -> prints the top of the rocprofv3 kernel stats
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/q_$TAG -o q -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/q_$TAG.log 2>&1
cd $REPO
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/q_$TAG/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.reader(open(f)))[:14]:
    print(r[0][:60], r[1:5])
PY
tail -1 $OUT/q_$TAG.log | cut -c1-400
