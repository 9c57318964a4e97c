#!/bin/bash
# round 6: cold operand panels touched up front (ORX_GEMM16_WARM: 0 never, 1 the top MLP's first product, 2 every nt launch) -- parity, per-launch medians, step
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6w; mkdir -p $O
echo skip > $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  for w in 0 1; do
    ORX_GEMM16_WARM=$w timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_w${w}_$rep.json 2>$O/err.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6w/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'], r.get('frac'))
    except Exception as e: print(f, 'ERR', e)
PY
for w in 0 1; do
cd /tmp
ORX_GEMM16_WARM=$w rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$w -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof_$w.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_$w -name "*kernel_trace.csv" | head -1); echo "== warm $w"; python scripts/step_positions.py $f | grep "gemm16_nt\|launches"
done
