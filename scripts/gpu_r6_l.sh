#!/bin/bash
# round 6, call l: closed-form replay in the row kernels (DLRM lazy Adam, sharded Adam, table flush) -- tests, DLRM Adam A/B
set -u
O=gpurun_out/r6l; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_rows_sorted.py tests/test_gpu_sharded.py tests/test_gpu_shard_engine.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_api.py tests/test_gpu_pairwise.py tests/test_gpu_fullsize.py tests/test_gpu_compose.py tests/test_gpu_reference_examples.py -q -m gpu -x -k "adam or Adam or lazy or checkpoint or resume or example" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
for v in cf loops cf loops; do
  if [ $v = loops ]; then export ORX_ADAM_NO_CF=1; else unset ORX_ADAM_NO_CF; fi
  timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --opt adam --steps 40 --warmup 10 > $O/dlrm_adam_$v.json 2> $O/dlrm_adam_$v.err
  python - <<P
import json
d=json.load(open("$O/dlrm_adam_$v.json")); print("dlrm adam $v", round(d["ms_per_step"]*1000,1), "us/step")
P
done
