#!/bin/bash
# round 6, call c: grouped dW + dX launches, loss folded into the head's backward, dense cast once per call -- tests, A/B, timeline
set -u
O=gpurun_out/r6c; mkdir -p $O
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_compose.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_reference_examples.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$n.json 2> $O/dlrm_$n.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$n.json")); print("$n", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
}
run all X=1
run nogroup ORX_GEMM16_NO_GROUP=1
run nofold ORX_DLRM_NO_FOLDED_LOSS=1
run castperstep ORX_DLRM_CAST_PER_STEP=1
run all2 X=1
run r5form ORX_GEMM16_NO_GROUP=1 ORX_DLRM_NO_FOLDED_LOSS=1 ORX_DLRM_CAST_PER_STEP=1 ORX_DLRM_NO_FUSED_SPARSE=1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T head_bwd_kernel > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-150
S=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats.csv
