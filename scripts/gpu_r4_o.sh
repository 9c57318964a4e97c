#!/bin/bash
# LDS-DMA staged fp16 products: correctness (DLRM suite) and A/B of the C5 fp16 step
rm -rf gpurun_out/r4o; mkdir -p gpurun_out/r4o
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_dlrm.py -q -x > gpurun_out/r4o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r4o/pytest.log
tail -5 gpurun_out/r4o/pytest.log
for d in 0 3 0 3; do
  ORX_GEMM16_DMA=$d timeout 200 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 2>/dev/null | tail -1 > gpurun_out/r4o/dlrm_dma${d}_$RANDOM.json
done
for f in gpurun_out/r4o/dlrm_dma*.json; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip() or '{}')
r=d.get('roofline',{})
print(sys.argv[1].split('/')[-1], 'ms/step', round(d.get('ms_per_step',0),4), 'gemm ms', round(r.get('gemm_ms_per_step',0),4), 'frac', round(r.get('frac',0),4))
PY
done
for i in 1 2; do ORX_GEMM16_NTS=0 timeout 200 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NTS=0 ms/step', round(d['ms_per_step'],4), 'gemm ms', round(d['roofline']['gemm_ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))"; done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4o/prof -o p -- python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 20 --warmup 5 > gpurun_out/r4o/prof.log 2>&1
f=$(find gpurun_out/r4o/prof -name '*kernel_trace.csv' | head -1)
python scripts/trace_by_shape.py "$f" > gpurun_out/r4o/shapes.txt; head -24 gpurun_out/r4o/shapes.txt
rm -rf gpurun_out/r4o/prof
