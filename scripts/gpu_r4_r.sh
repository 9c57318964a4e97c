#!/bin/bash
rm -rf gpurun_out/r4r; mkdir -p gpurun_out/r4r
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_reference_examples.py -q -x > gpurun_out/r4r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4r/pytest.log
tail -4 gpurun_out/r4r/pytest.log
one() { timeout 200 python bench.py --no-cpu-baseline --model dlrm "$@" --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],4), 'gemm ms', round(d['roofline'].get('gemm_ms_per_step',0),4), 'frac', round(d['roofline']['frac'],4))"; }
for i in 1 2; do
  echo "side apply on:"; one --fp16-mlp
  echo "side apply off:"; ORX_DLRM_NO_SIDE_APPLY=1 one --fp16-mlp
done
echo "fp32 side on:"; one
echo "fp32 side off:"; ORX_DLRM_NO_SIDE_APPLY=1 one
echo "adagrad fp16 side on:"; one --fp16-mlp --opt adagrad
echo "adagrad fp16 side off:"; ORX_DLRM_NO_SIDE_APPLY=1 one --fp16-mlp --opt adagrad
