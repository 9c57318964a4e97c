#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py -m gpu -q -x --timeout 600 > gpurun_out/r3g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3g_pytest.log
for v in "" "ORX_DLRM_NO_DW_STREAM=1" "ORX_INTERACT_NO_SPLIT=1" "ORX_DLRM_NO_DW_STREAM=1 ORX_INTERACT_NO_SPLIT=1"; do
  tag=$(echo "$v" | tr -c 'A-Z0-9_\n' '_'); [ -z "$tag" ] && tag=default
  env $v timeout 300 python bench.py --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3g_dlrm_$tag.json 2>> gpurun_out/r3g.err
done
timeout 300 python bench.py --model dlrm --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3g_dlrm_fp32.json 2>> gpurun_out/r3g.err
tail -n 4 gpurun_out/r3g_pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3g_dlrm_*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline',{})
        print(f, 'ms/step', round(d['ms_per_step'],4), 'gemm_ms', round(r.get('gemm_ms_per_step',0),4), 'TF', round(r.get('achieved',0),1))
    except Exception as e:
        print(f, 'ERR', e)
PY
