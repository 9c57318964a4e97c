#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -12
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), 'frac', round(d['roofline']['frac'],3))"; }
for i in 1 2 3; do echo -n "sgd K=20: "; one --steps 20 --warmup 5; done
ORX_HOST_TIMING=1 timeout 200 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>&1 >/dev/null | grep "orx host"
