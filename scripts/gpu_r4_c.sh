#!/bin/bash
mkdir -p gpurun_out/r4c
for k in 20 200; do
ORX_PLAN_TIMING=1 timeout 300 python bench.py --steps $k --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "plan timing" | tail -2
ORX_NO_PAIR=1 ORX_PLAN_TIMING=1 timeout 300 python bench.py --steps $k --warmup 5 --no-secondary --no-cpu-baseline 2>&1 >/dev/null | grep "plan timing" | tail -2
done
