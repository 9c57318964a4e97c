#!/bin/bash
# round 5, call p: GMF's dense-gradient partials per workgroup instead of per wavefront -- tests, bench
set -u
O=gpurun_out/r5p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_compose.py tests/test_gpu_reference_examples.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
  timeout 300 python bench.py --no-cpu-baseline --model gmf --steps 200 --warmup 20 > $O/gmf_$i.json 2> $O/gmf_$i.err
  python - <<P
import json
d=json.load(open("$O/gmf_$i.json")); print("gmf", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(d["roofline"]["kernel_us"],2))
P
done
