#!/bin/bash
rm -rf gpurun_out/r4s; mkdir -p gpurun_out/r4s
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_sharded_dlrm.py -q -x > gpurun_out/r4s/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4s/pytest.log
tail -3 gpurun_out/r4s/pytest.log
one() { timeout 200 python bench.py --no-cpu-baseline --model dlrm "$@" --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4), 'gemm ms', round(d['roofline'].get('gemm_ms_per_step',0),4), 'frac', round(d['roofline']['frac'],4))"; }
for i in 1 2; do
  for n in 0 2 4 8; do echo "PIPE=$n:"; ORX_INTERACT_PIPE=$n one --fp16-mlp; done
done
