#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for dbg in 0 1; do
export ORX_RANK_DBG=$dbg
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$dbg -- python $R/scripts/bench_score.py --evaluate --pos 10 > /dev/null 2> /tmp/err_$dbg.log
f=$(find /tmp/prof_$dbg -name '*kernel_stats.csv' | head -n 1)
echo "dbg=$dbg"; grep rank_metrics_csr "$f"
done
