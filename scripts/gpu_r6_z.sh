#!/bin/bash
# round 6: the sorted apply's finish pass on a side stream beside the dense optimizer launch (ORX_DLRM_SIDE_FINISH=1), A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6z; mkdir -p $O
ORX_DLRM_SIDE_FINISH=1 timeout 900 python -m pytest tests/test_gpu_dlrm.py -x -q -m gpu -k "not bit_identical" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_base_$rep.json 2>$O/err.log
  ORX_DLRM_SIDE_FINISH=1 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_side_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6z/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
