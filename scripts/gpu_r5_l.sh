#!/bin/bash
# round 5, call l: step results and plan counters stored through the bus by the last kernel (no copy commands behind it) -- A/B against the build before
# (the experiment's sources were not kept: profiles/r5_plan_levers.txt.  The recipe for any two-build A/B: build the other tree, copy its
#  openrec_amd/_lib/libopenrec_hip.so to scratch/ab/libopenrec_hip_old.so -- *.so files travel with gpurun -- and select it with ORX_LIB_PATH)
set -u
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_pairing.py tests/test_gpu_edges.py tests/test_gpu_pointwise.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_stepqueue.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export ORX_LIB_PATH=$(pwd)/scratch/ab/libopenrec_hip_old.so; else unset ORX_LIB_PATH; fi
    timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/k20_${v}_$i.json 2> $O/k20_${v}_$i.err
    python - <<P
import json
d=json.load(open("$O/k20_${v}_$i.json")); print("$v k20", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(d["roofline"]["kernel_us"],2))
P
  done
done
unset ORX_LIB_PATH
