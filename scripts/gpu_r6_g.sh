#!/bin/bash
# round 6, call g: the top MLP's weight-gradient products on a side stream beside the interaction backward
set -u
O=gpurun_out/r6g; mkdir -p $O
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_compose.py tests/test_gpu_sharded_dlrm.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$n.json 2> $O/dlrm_$n.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$n.json")); print("$n", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
}
run defer1 X=1
run defer0 ORX_DLRM_DEFER_DW=0
run defer2 ORX_DLRM_DEFER_DW=2
run defer1b X=1
run defer0b ORX_DLRM_DEFER_DW=0
run defer2b ORX_DLRM_DEFER_DW=2
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T head_bwd_kernel > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-150
