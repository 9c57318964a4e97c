#!/bin/bash
# round 6, call p: the NT epilogue's operands requested before the loop tail -- the decomposition again, DLRM tests, bench, timeline
set -u
O=gpurun_out/r6p; mkdir -p $O
REPO=$(pwd)
(cd scratch && timeout 300 ./exp_k512 > ../$O/exp_k512_after.txt 2>&1); grep -E "^==|forward form|backward form \(|epilogue only|no epilogue  " $O/exp_k512_after.txt
timeout 1800 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_compose.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_modules.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$i.json 2> $O/dlrm_$i.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$i.json")); print("run $i", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1), "frac", round(d["roofline"]["frac"],3))
P
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T head_bwd_kernel > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-150
