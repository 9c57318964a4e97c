#!/bin/bash
# round 6: the in-launch hand-offs with one-word polling first -- GMF reducers and csr finishers; parity, then A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6t; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rows_sorted.py tests/test_gpu_dlrm.py tests/test_gpu_pointwise.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_fold_$rep.json 2>$O/err.log
  ORX_CSR_TWO_LAUNCHES=1 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_two_$rep.json 2>$O/err.log
  timeout 300 python bench.py --model gmf --no-cpu-baseline --steps 200 --warmup 20 > $O/gmf_tail_$rep.json 2>$O/err.log
  ORX_POINT_NO_WTAIL=1 timeout 300 python bench.py --model gmf --no-cpu-baseline --steps 200 --warmup 20 > $O/gmf_launches_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6t/*_?.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], r.get('frac'), r.get('kernel_us'))
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-140 {} | grep -i "csr\|interact\|dense_apply"'
