#!/bin/bash
# round 6: csr_finish folded into csr_apply's launch (finisher workgroups) -- parity of everything on the sorted apply, then the C5 step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rows_sorted.py tests/test_gpu_dlrm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_fold_$rep.json 2>$O/err.log
  ORX_CSR_TWO_LAUNCHES=1 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_two_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6s/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], r.get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -c1-140 {} | head -24'
