#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for g in 0 16; do
ORX_ADAM_PREPASS=$g timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a$g -- python $R/bench.py --opt adam --steps 128 --warmup 64 --no-cpu-baseline > /dev/null 2> /tmp/err.log
f=$(find /tmp/prof_a$g -name '*kernel_stats.csv' | head -n 1)
echo "G=$g"; head -n 8 "$f" | cut -c1-140
done
