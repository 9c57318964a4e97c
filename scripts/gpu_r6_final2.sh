#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r6_pytest_gpu_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6_pytest_gpu_all.log
tail -4 gpurun_out/r6_pytest_gpu_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6_smoke.log 2>&1; tail -2 gpurun_out/r6_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_bench_driver_protocol.json 2> gpurun_out/r6_bench_driver_protocol.err; cut -c1-900 gpurun_out/r6_bench_driver_protocol.json
timeout 600 python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; cut -c1-600 gpurun_out/r6_bench_default.json
