#!/bin/bash
# round 6, call e: slices policy (16 for 5..16 tiles, masked slab rows), the 128 x 64 wave tile for the 256 x 128 products
set -u
O=gpurun_out/r6e; mkdir -p $O
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
ORX_GEMM16_WAVE_TILE=128 timeout 1500 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py -q -m gpu -x > $O/tests_wt128.log 2>&1; echo "tests wt128 rc=$?"; tail -3 $O/tests_wt128.log
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$n.json 2> $O/dlrm_$n.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$n.json")); print("$n", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
}
run base X=1
run wt128 ORX_GEMM16_WAVE_TILE=128
run base2 X=1
run wt128b ORX_GEMM16_WAVE_TILE=128
(cd /tmp && export TMPDIR=/tmp && ORX_GEMM16_WAVE_TILE=128 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T head_bwd_kernel > $O/timeline_wt128.txt 2>&1; cat $O/timeline_wt128.txt | cut -c1-150
