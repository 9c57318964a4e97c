#!/bin/bash
rm -rf gpurun_out/r4t; mkdir -p gpurun_out/r4t
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step', round(d['ms_per_step']*1e3,2), 'us  kernel', round(r.get('kernel_us',0),2), 'frac', round(r['frac'],4))"; }
for i in 1 2 3; do
  echo "K=200 plain:"; one --steps 200 --warmup 20
  echo "K=200 write-through:"; ORX_ROWS_WT=1 one --steps 200 --warmup 20
  echo "K=20 plain:"; one --steps 20 --warmup 5
  echo "K=20 write-through:"; ORX_ROWS_WT=1 one --steps 20 --warmup 5
done
