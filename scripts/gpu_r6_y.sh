#!/bin/bash
# experiment: the all-item scorer with a store instruction covering 4 rows x 256 B instead of 16 rows x 64 B (wrong layout on purpose, same bytes)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6y; mkdir -p $O
for rep in 1 2; do
python scripts/bench_score.py 2>/dev/null | tail -1 > $O/score_normal_$rep.json
ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/libopenrec_exp_score.so python scripts/bench_score.py 2>/dev/null | tail -1 > $O/score_exp_$rep.json
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6y/score_*.json')):
    try:
        d=json.loads(open(f).read()); print(f.split('/')[-1], 'kernel_us %.1f' % d['kernel_us'], 'mfma TF %.1f' % d['mfma_tflops'])
    except Exception as e: print(f, 'ERR', e)
PY
