#!/bin/bash
# round 5, measurement A: where the K = 20 driver-protocol call spends its time WITHOUT a tracing tool (dispatch-attached events in
# launch order: ORX_PROF_TIMELINE; the host's way through the call: ORX_HOST_TIMING)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_a; mkdir -p $O
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>$O/err.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), d['roofline'].get('other_kernels_us'))"; }
for i in 1 2 3; do echo -n "K=20: "; one --steps 20 --warmup 5; done
echo -n "K=200: "; one --steps 200 --warmup 5
ORX_PROF_TIMELINE=1 ORX_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/timeline.out 2> $O/timeline.err
grep "orx host" $O/timeline.err
grep "orx timeline" $O/timeline.err | head -60
ORX_PLAN_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>&1 | grep "plan timing" | tail -2
