#!/bin/bash
# round 3, call q: lazy Adam with the long-gap pre-pass
cd /root/repo
mkdir -p gpurun_out
for g in 0 8 16 32; do
echo "ORX_ADAM_PREPASS=$g"
ORX_ADAM_PREPASS=$g timeout 600 python bench.py --opt adam --steps 128 --warmup 64 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3q_adam_pre$g.json | grep -o '"ms_per_step": [0-9.]*\|"kernel_us": [0-9.]*\|"dup_apply": [0-9.]*'
done
ORX_ADAM_PREPASS=16 timeout 1500 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -n 3
