#!/bin/bash
rm -rf gpurun_out/r4s; mkdir -p gpurun_out/r4s
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_reference_examples.py tests/test_gpu_compose.py -q -x 2>&1 | tail -2
one() { timeout 200 python bench.py --no-cpu-baseline --model dlrm "$@" --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4), 'gemm ms', round(d['roofline'].get('gemm_ms_per_step',0),4), 'frac', round(d['roofline']['frac'],4))"; }
for i in 1 2 3; do one --fp16-mlp; done
one
