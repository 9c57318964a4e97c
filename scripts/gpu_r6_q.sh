#!/bin/bash
# round 6: GMF's Dense(1) gradient reduced inside the step's launch (dense_tail) -- parity, then A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_pairing.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model gmf --no-cpu-baseline --steps 200 --warmup 20 > $O/gmf_tail_$rep.json 2>$O/err.log
  ORX_POINT_NO_WTAIL=1 timeout 300 python bench.py --model gmf --no-cpu-baseline --steps 200 --warmup 20 > $O/gmf_launches_$rep.json 2>$O/err.log
done
for opt in adagrad adam; do
  timeout 300 python bench.py --model gmf --no-cpu-baseline --opt $opt --steps 200 --warmup 20 > $O/gmf_${opt}_tail.json 2>$O/err.log
  ORX_POINT_NO_WTAIL=1 timeout 300 python bench.py --model gmf --no-cpu-baseline --opt $opt --steps 200 --warmup 20 > $O/gmf_${opt}_launches.json 2>$O/err.log
done
timeout 300 python bench.py --model gmf --no-cpu-baseline --steps 20 --warmup 5 > $O/gmf_k20_tail.json 2>$O/err.log
ORX_POINT_NO_WTAIL=1 timeout 300 python bench.py --model gmf --no-cpu-baseline --steps 20 --warmup 5 > $O/gmf_k20_launches.json 2>$O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6q/gmf_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], 'kernel_us %.2f' % r['kernel_us'], r.get('other_kernels_us'))
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model gmf --no-cpu-baseline --steps 200 --warmup 20 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/prof/*kernel_stats.csv $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); head -6 "$f" | cut -c1-160
