#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_rccl_rank1.py -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --gpus 1 --sharded --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sharded world 1: ms/step', round(d['ms_per_step'],5), d.get('phases_us'))"; done
