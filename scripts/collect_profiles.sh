#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers rest on (run on the GPU box from the repo root):
#   scripts/collect_profiles.sh <tag>      ->  gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
# Three separate rocprofv3 runs: --kernel-trace --stats, then one --pmc pass per counter (never combined with
# other trace domains), then the un-profiled bench line.
set -u
TAG=${1:-rX}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline"
BENCH_SHORT="python $REPO/bench.py --steps 40 --warmup 5 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- $BENCH > $OUT/${TAG}_prof_run.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o f -- $BENCH_SHORT > $OUT/${TAG}_pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o w -- $BENCH_SHORT > $OUT/${TAG}_pmc_write.log 2>&1
cd $REPO
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
python scripts/summarize_profiles.py $TAG
