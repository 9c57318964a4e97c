#!/bin/bash
# round 6: lazily applied Adam -- the step of the rows referenced once taken inside the interaction backward (ORX_DLRM_NO_FUSED_ADAM=1: off); parity, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6ae; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py tests/test_gpu_sharded_dlrm.py tests/test_gpu_reference_examples.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --opt adam --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_adam_fused_$rep.json 2>$O/err.log
  ORX_DLRM_NO_FUSED_ADAM=1 timeout 300 python bench.py --model dlrm --fp16-mlp --opt adam --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_adam_sorted_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6ae/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --opt adam --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_positions.py $f | grep "interact_bwd\|csr\|dense_apply\|launches"
