#!/bin/bash
# experiments: 16 split-K slices in flight per thread of dense_apply_fused (lib_un16.so); act_bwd_colsum with slabs of 64 rows (ORX_EXP_ACT_SLAB)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6ac; mkdir -p $O
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_base_$rep.json 2>$O/err.log
  ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/libs/lib_un16.so timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_un16_$rep.json 2>$O/err.log
  ORX_EXP_ACT_SLAB=64 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_slab64_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6ac/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
