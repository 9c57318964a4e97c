#!/bin/bash
# round 3, call o: the whole GPU suite, then the DLRM modules five more times, the smoke entry and the default bench line
cd /root/repo
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > gpurun_out/r3o_pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3o_pytest_all.log
tail -n 6 gpurun_out/r3o_pytest_all.log
for i in 1 2 3 4 5; do timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_sharded_dlrm.py -q -x 2>&1 | tail -n 1; done | tee gpurun_out/r3o_dlrm_repeat.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
( time python bench.py ) > gpurun_out/r3o_bench_default.json 2> gpurun_out/r3o_bench_default.err; tail -n 4 gpurun_out/r3o_bench_default.err; cut -c1-600 gpurun_out/r3o_bench_default.json
