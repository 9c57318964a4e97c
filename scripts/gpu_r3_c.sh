#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export ORX_TEST_RECORD=$PWD/gpurun_out/r3c_records.jsonl
rm -f "$ORX_TEST_RECORD"
( time timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --durations=15 ) > gpurun_out/r3c_pytest_all.log 2>&1
echo "pytest_all rc=$?" >> gpurun_out/r3c_pytest_all.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r3c_bench.json 2> gpurun_out/r3c_bench.err
tail -n 30 gpurun_out/r3c_pytest_all.log
cat gpurun_out/r3c_bench.json; tail -n 5 gpurun_out/r3c_bench.err
