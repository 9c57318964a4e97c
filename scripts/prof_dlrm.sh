#!/bin/bash
# kernel-level profile of the DLRM fp16 step: scripts/prof_dlrm.sh <tag> [extra bench args]  ->  gpurun_out/<tag>_dlrm_stats.txt
TAG=${1:-d}; shift
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 "$@" > $OUT/${TAG}_prof.log 2>&1
cd $REPO
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof_$TAG/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = open("$OUT/${TAG}_dlrm_stats.txt", "w")
for r in rows[:26]:
    line = f"{r['Name'][:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {float(r['TotalDurationNs'])/1e3/50:8.1f} us  min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}"
    print(line); out.write(line + "\n")
PY
