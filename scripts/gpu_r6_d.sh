#!/bin/bash
# round 6, call d: split-K slices of the small gradients (8 by default; 32 = round 5), bit-identity of the grouped launch
set -u
O=gpurun_out/r6d; mkdir -p $O
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log; grep -n "AssertionError" $O/tests.log | head
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$n.json 2> $O/dlrm_$n.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$n.json")); print("$n", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
}
run s8 X=1
run s32 ORX_GEMM16_TN_SMALL_S=32
run s4 ORX_GEMM16_TN_SMALL_S=4
run s16 ORX_GEMM16_TN_SMALL_S=16
run s8b X=1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T head_bwd_kernel > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-150
