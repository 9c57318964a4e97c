#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 120 python scripts/dbg_r5_skew.py default 2>&1 | grep -v amdgpu.ids
ORX_PLAN_WAIT=1 timeout 120 python scripts/dbg_r5_skew.py wait 2>&1 | grep -v amdgpu.ids
ORX_NO_PAIR=1 timeout 120 python scripts/dbg_r5_skew.py nopair 2>&1 | grep -v amdgpu.ids
ORX_FORCE_FALLBACK=2 timeout 120 python scripts/dbg_r5_skew.py noinline 2>&1 | grep -v amdgpu.ids
ORX_FORCE_FALLBACK=8 ORX_PLAN_WAIT=1 timeout 120 python scripts/dbg_r5_skew.py nostaging_wait 2>&1 | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_gpu_modules.py -x -q -m gpu 2>&1 | tail -5
