#!/bin/bash
# round 6, call o: the weight-gradient kernel with two K groups per workgroup (eight wavefronts) -- tests, A/B on one box, timeline
set -u
O=gpurun_out/r6o; mkdir -p $O
REPO=$(pwd)
timeout 1800 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_compose.py tests/test_gpu_sharded_dlrm.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$n.json 2> $O/dlrm_$n.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$n.json")); print("$n", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
}
run kg2 X=1
run tn3 ORX_GEMM16_TN_DMA=3
run kg2b X=1
run tn3b ORX_GEMM16_TN_DMA=3
run kg2c X=1
run tn3c ORX_GEMM16_TN_DMA=3
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T head_bwd_kernel > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-150
