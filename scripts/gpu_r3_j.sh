#!/bin/bash
# round 3, call j: CSR rank metrics
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_metrics.py -x -q > gpurun_out/r3j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3j_pytest.log
tail -n 5 gpurun_out/r3j_pytest.log
timeout 600 python scripts/bench_score.py --evaluate > gpurun_out/r3j_eval_1000x1M.json 2> gpurun_out/r3j_eval.err; grep -o '"evaluate_ms": [0-9.]*' gpurun_out/r3j_eval_1000x1M.json
timeout 600 python scripts/bench_score.py --evaluate --users 1000 --items 100000 > gpurun_out/r3j_eval_1000x100k.json 2>> gpurun_out/r3j_eval.err;  grep -o '"evaluate_ms[a-z_]*": [0-9.]*' gpurun_out/r3j_eval_1000x100k.json
timeout 600 python scripts/bench_score.py --evaluate --pos 40 > gpurun_out/r3j_eval_pos40.json 2>> gpurun_out/r3j_eval.err;  grep -o '"evaluate_ms": [0-9.]*' gpurun_out/r3j_eval_pos40.json
