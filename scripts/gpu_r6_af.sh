#!/bin/bash
# round 6: interact_fwd with d as a template constant -- the sample's rows all requested before the first product (ORX_INTERACT_FWD_RT=1: the runtime-d loop)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6af; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dlrm.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_dt_$rep.json 2>$O/err.log
  ORX_INTERACT_FWD_RT=1 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_rt_$rep.json 2>$O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6af/dlrm_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'])
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_positions.py $f | grep "interact\|launches"
