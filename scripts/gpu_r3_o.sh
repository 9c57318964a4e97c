#!/bin/bash
# round 3, call o: the whole GPU suite, the smoke entry and the default bench line (final build of the round)
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r3o_pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3o_pytest_all.log
tail -n 6 gpurun_out/r3o_pytest_all.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
( time python bench.py ) > gpurun_out/r3o_bench_default.json 2> gpurun_out/r3o_bench_default.err; tail -n 4 gpurun_out/r3o_bench_default.err; cut -c1-300 gpurun_out/r3o_bench_default.json
