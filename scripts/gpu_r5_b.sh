#!/bin/bash
# round 5, B: the bias experiment (scratch/exp_bias.hip) and the no-read-back call (ORX_PLAN_WAIT=1: the old form) A/B on one box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_b; mkdir -p $O
timeout 120 scratch/exp_bias | tee $O/exp_bias.txt
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), d['roofline'].get('other_kernels_us'))"; }
for i in 1 2 3; do echo -n "K=20 nowait: "; one --steps 20 --warmup 5; echo -n "K=20 wait:   "; ORX_PLAN_WAIT=1 one --steps 20 --warmup 5; done
echo -n "K=200 nowait: "; one --steps 200 --warmup 5
echo -n "K=200 wait:   "; ORX_PLAN_WAIT=1 one --steps 200 --warmup 5
ORX_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>&1 >/dev/null | grep "orx host"
timeout 900 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_stress.py tests/test_gpu_stepqueue.py -x -q -m gpu 2>&1 | tail -5
