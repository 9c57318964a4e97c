#!/bin/bash
# Collects the rocprofv3 evidence behind every number DESIGN.md quotes (run on the GPU box from the repo root):
#   scripts/collect_profiles.sh <tag> [all|head]   ->  gpurun_out/<tag>_*   (copy what is to be judged into profiles/)
# Per workload: one `rocprofv3 --kernel-trace --stats` run (kernel stats CSV) and the un-profiled bench line; for the
# HBM-bound fused kernels additionally one --pmc pass per counter (FETCH_SIZE, WRITE_SIZE; never combined with other
# trace domains).  scripts/summarize_profiles.py boils the output down and stamps the traffic file with the source hash.
set -u
TAG=${1:-rX}
WHAT=${2:-all}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp

run() {   # run <name> <pmc 0|1> <bench args...>
  local NAME=$1 PMC=$2; shift 2
  local CMD="python $REPO/bench.py --no-cpu-baseline $*"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_$NAME -o b -- $CMD > $OUT/${TAG}_${NAME}_prof.log 2>&1
  if [ "$PMC" = "1" ]; then
    local SHORT=$(echo "$CMD" | sed -E 's/--steps [0-9]+/--steps 40/; s/--warmup [0-9]+/--warmup 5/')
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmcf_${TAG}_$NAME -o f -- $SHORT > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmcw_${TAG}_$NAME -o w -- $SHORT > /dev/null 2>&1
  fi
  (cd $REPO && timeout 600 python bench.py --no-cpu-baseline "$@" > $OUT/${TAG}_${NAME}_bench.json 2> /dev/null)
}

# the headline (configs[1]): the driver's protocol (20 steps) and the steady state (200 steps), with the CPU baseline once
run c2 1 --steps 200 --warmup 20
(cd $REPO && timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_c2_k20_bench.json 2> /dev/null)
if [ "$WHAT" = "all" ]; then
  run c3_ucml128_censor 1 --model ucml --dim 128 --censor --steps 200 --warmup 20
  run bpr_adagrad 1 --opt adagrad --steps 200 --warmup 20
  run bpr_adam 0 --opt adam --steps 128 --warmup 64
  run bpr_zipf 0 --zipf 1.05 --steps 200 --warmup 20
  run wrmf 1 --model wrmf --steps 200 --warmup 20
  run gmf 0 --model gmf --steps 200 --warmup 20
  run dlrm_fp16 0 --model dlrm --fp16-mlp --steps 40 --warmup 10
  run dlrm_fp32 0 --model dlrm --steps 40 --warmup 10
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_score -o b -- python $REPO/scripts/bench_score.py > $OUT/${TAG}_score_prof.log 2>&1
  (cd $REPO && python scripts/bench_score.py > $OUT/${TAG}_score_bench.json 2> /dev/null)
fi
cd $REPO
if [ "$WHAT" = "all" ]; then
  bash scripts/pmc_dlrm.sh $TAG > /dev/null 2>&1            # MFMA-busy / LDS-conflict counters of the DLRM fp16 step -> <tag>_dlrm_pmc.csv
  [ -x scratch/copy_bw ] && (cd scratch && timeout 120 ./copy_bw > $OUT/${TAG}_copy_bw.log 2>&1)   # the streaming-copy yardstick
fi
python scripts/summarize_profiles.py $TAG
