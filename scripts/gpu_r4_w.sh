#!/bin/bash
# launch-path environment switches of the HIP runtime on the driver-protocol call (no library change)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { timeout 100 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2))"; }
for i in 1 2 3; do
  echo -n "default: "; one
  echo -n "HIP_FORCE_DEV_KERNARG=1: "; HIP_FORCE_DEV_KERNARG=1 one
  echo -n "HIP_FORCE_DEV_KERNARG=0: "; HIP_FORCE_DEV_KERNARG=0 one
done
