#!/bin/bash
# round 6: XCD-aware sample mapping in the interaction kernels (a contiguous eighth of the batch per XCD, as the products map their row tiles)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dlrm.py tests/test_gpu_c5_shapes.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_xcd_$rep.json 2>$O/err.log
  ORX_INTERACT_NO_XCD=1 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_rr_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6u/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], r.get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_xcd -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
ORX_INTERACT_NO_XCD=1 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_rr -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof2.log 2>&1
cd $GRAFT_REPO_ROOT
for v in xcd rr; do f=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); echo "== $v"; python scripts/step_positions.py $f; done
