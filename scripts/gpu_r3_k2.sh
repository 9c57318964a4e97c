#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f -- python $R/scripts/bench_score.py --evaluate --pos 10 > /tmp/f.json 2> /tmp/err_f.log
f=$(find /tmp/prof_f -name '*kernel_stats.csv' | head -n 1)
grep "ScoreArgs\|rank_fused\|rank_finish\|mask_bits\|gather" "$f" | cut -c1-150
cd $R; timeout 600 python -m pytest tests/test_metrics.py -x -q 2>&1 | tail -n 3
timeout 600 python scripts/bench_score.py --evaluate 2>/dev/null | grep -o '"evaluate_ms": [0-9.]*'
