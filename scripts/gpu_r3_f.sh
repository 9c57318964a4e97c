#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_rows_sorted.py -m gpu -q -x --timeout 600 > gpurun_out/r3f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r3f_pytest.log
timeout 300 python bench.py --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3f_dlrm_fp16.json 2> gpurun_out/r3f_dlrm_fp16.err
ORX_DLRM_NO_FUSED_DENSE=1 timeout 300 python bench.py --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3f_dlrm_fp16_nofuse.json 2>> gpurun_out/r3f_dlrm_fp16.err
timeout 300 python bench.py --model dlrm --fp16-mlp --opt adagrad --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3f_dlrm_fp16_adagrad.json 2>> gpurun_out/r3f_dlrm_fp16.err
timeout 300 python bench.py --model dlrm --fp16-mlp --opt adam --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r3f_dlrm_fp16_adam.json 2>> gpurun_out/r3f_dlrm_fp16.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3f_prof_dlrm -o dlrm -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --steps 40 --warmup 10 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3f_prof_dlrm.log 2>&1
cd $GRAFT_REPO_ROOT
tail -n 5 gpurun_out/r3f_pytest.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f_dlrm_*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline',{})
        print(f, 'ms/step', round(d['ms_per_step'],4), 'gemm_ms', round(r.get('gemm_ms_per_step',0),4), 'TF', round(r.get('achieved',0),1))
    except Exception as e:
        print(f, 'ERR', e)
PY
find gpurun_out/r3f_prof_dlrm -name "*kernel_stats.csv" | head -2
