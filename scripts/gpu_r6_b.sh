#!/bin/bash
# round 6, call b: rows referenced once applied inside interact_bwd -- DLRM tests, A/B bench on one box, kernel profile + step timeline
set -u
O=gpurun_out/r6b; mkdir -p $O
REPO=$(pwd)
timeout 1200 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_compose.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_rows_sorted.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
for v in fused sorted fused sorted; do
  if [ $v = sorted ]; then export ORX_DLRM_NO_FUSED_SPARSE=1; else unset ORX_DLRM_NO_FUSED_SPARSE; fi
  timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$v.json 2> $O/dlrm_$v.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$v.json")); print("$v", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
done
unset ORX_DLRM_NO_FUSED_SPARSE
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $REPO/$O/prof.log 2>&1)
T=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python scripts/step_timeline.py $T dlrm_loss_kernel > $O/timeline.txt 2>&1; cat $O/timeline.txt
S=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats.csv; head -30 $S | cut -c1-150
