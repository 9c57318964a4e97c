#!/bin/bash
# round 5, call o: the sharded SGD step's two applies (duplicated user rows + item rows) in one launch -- tests, A/B (ORX_APPLY_NO_PAIR=1: one by one)
set -u
O=gpurun_out/r5o; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_shard_engine.py tests/test_gpu_sharded.py tests/test_gpu_rccl_rank1.py tests/test_gpu_rccl_multirank.py tests/test_gpu_rows_sorted.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1 2 3; do
  for v in pair single; do
    if [ $v = single ]; then export ORX_APPLY_NO_PAIR=1; else unset ORX_APPLY_NO_PAIR; fi
    timeout 300 python bench.py --no-cpu-baseline --sharded --steps 128 --warmup 64 > $O/sh_${v}_$i.json 2> $O/sh_${v}_$i.err
    python - <<P
import json
d=json.load(open("$O/sh_${v}_$i.json")); print("$v:", round(d["ms_per_step"]*1000,2), "us/step", d.get("phases_us"))
P
  done
done
unset ORX_APPLY_NO_PAIR
