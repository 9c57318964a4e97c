#!/bin/bash
# round 3, call A: the deterministic row apply + the DLRM parity rewrite, then the DLRM step A/B (sorted apply vs round-2 atomics)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export ORX_TEST_RECORD=$PWD/gpurun_out/r3a_records.jsonl
rm -f "$ORX_TEST_RECORD"
timeout 1500 python -m pytest tests/test_gpu_rows_sorted.py tests/test_gpu_dlrm.py tests/test_gpu_sampler.py tests/test_gpu_sharded_dlrm.py -m gpu -q --timeout 600 > gpurun_out/r3a_pytest1.log 2>&1
echo "pytest1 rc=$?" >> gpurun_out/r3a_pytest1.log
timeout 900 python -m pytest tests/test_gpu_c5_shapes.py -m gpu -q --timeout 800 > gpurun_out/r3a_pytest_c5.log 2>&1
echo "pytest_c5 rc=$?" >> gpurun_out/r3a_pytest_c5.log
for mode in sorted atomics; do
  for fp in "--fp16-mlp" ""; do
    if [ $mode = atomics ]; then export ORX_ROWS_ATOMICS=1; else unset ORX_ROWS_ATOMICS; fi
    timeout 300 python bench.py --model dlrm $fp --steps 40 --warmup 10 --no-cpu-baseline > "gpurun_out/r3a_dlrm_${mode}${fp// /}.json" 2> "gpurun_out/r3a_dlrm_${mode}${fp// /}.err"
  done
done
unset ORX_ROWS_ATOMICS
tail -n 5 gpurun_out/r3a_pytest1.log gpurun_out/r3a_pytest_c5.log
cat gpurun_out/r3a_dlrm_*.json
