#!/bin/bash
# round 3, call k: kernel times of the evaluation step
cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r3k
for cfg in "1000 10" "1000 40" "64 10" "256 100"; do
set -- $cfg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$1_$2 -- python $R/scripts/bench_score.py --evaluate --users $1 --pos $2 > $R/gpurun_out/r3k/eval_u$1_pos$2.json 2> /tmp/err.log
f=$(find /tmp/prof_$1_$2 -name '*kernel_stats.csv' | head -n 1)
cp "$f" $R/gpurun_out/r3k/eval_u$1_pos$2_kernel_stats.csv
echo "users $1 pos $2"; grep -o '"evaluate_ms": [0-9.]*' $R/gpurun_out/r3k/eval_u$1_pos$2.json; grep "rank_\|score_mfma\|mask_bits" "$f"
done
