#!/bin/bash
# the C5 step with the example's optimizer (Adam, applied lazily to the embedding rows): launch by launch
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6ad; mkdir -p $O
timeout 300 python bench.py --model dlrm --fp16-mlp --opt adam --no-cpu-baseline --steps 50 --warmup 10 > $O/dlrm_adam.json 2>$O/err.log
python -c "
import json; d=json.loads(open('$O/dlrm_adam.json').read().strip().splitlines()[-1]); print('ms/step', d['ms_per_step'])"
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --opt adam --no-cpu-baseline --steps 50 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_trace.csv" | head -1); python scripts/step_positions.py $f
