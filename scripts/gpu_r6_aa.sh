#!/bin/bash
# round 6: launches folded away -- the finish pass carried by the dense optimizer launch (ORX_DLRM_FINISH_LAUNCH=1: own launch), the head forward inside the head backward (ORX_DLRM_HEAD_FWD_LAUNCH=1: own launch); parity, then A/B of the latter on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6aa; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_dlrm.py  -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for rep in 1 2 3; do
  timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_carried_$rep.json 2>$O/err.log
  ORX_DLRM_HEAD_FWD_LAUNCH=1 timeout 300 python bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_launch_$rep.json 2>$O/err.log
done
timeout 300 python bench.py --model dlrm --fp16-mlp --opt adagrad --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_adagrad_carried.json 2>$O/err.log
ORX_DLRM_HEAD_FWD_LAUNCH=1 timeout 300 python bench.py --model dlrm --fp16-mlp --opt adagrad --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_adagrad_launch.json 2>$O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6aa/dlrm_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'ms/step %.5f' % d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_positions.py $f | sed -n 1,3p; python scripts/step_positions.py $f | tail -1
