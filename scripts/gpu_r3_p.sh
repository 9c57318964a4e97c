#!/bin/bash
# round 3, call p: GMF's Dense(1) kernel updated inside the step launches
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_api.py tests/test_gpu_compose.py tests/test_gpu_reference_examples.py tests/test_gpu_stepqueue.py -x -q 2>&1 | tail -n 8
for v in 0 1; do
echo "ORX_GMF_DENSE_LAUNCHES=$v"
ORX_GMF_DENSE_LAUNCHES=$v timeout 600 python bench.py --model gmf --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3p_gmf_$v.json | grep -o '"ms_per_step": [0-9.]*\|"kernel_us": [0-9.]*'
ORX_GMF_DENSE_LAUNCHES=$v timeout 600 python bench.py --model gmf --opt adagrad --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3p_gmf_adagrad_$v.json | grep -o '"ms_per_step": [0-9.]*'
done
