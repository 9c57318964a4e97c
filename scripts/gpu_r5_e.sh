#!/bin/bash
# Adagrad with pairing (new build) against the build before it (scratch/lib_base.so) on one box; then the tests that cover it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), 'frac', round(d['roofline']['frac'],3), d['roofline'].get('other_kernels_us'))"; }
for i in 1 2 3; do
  echo -n "adagrad K=20 pairs:      "; one --opt adagrad --steps 20 --warmup 5
  echo -n "adagrad K=20 base build: "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_base.so one --opt adagrad --steps 20 --warmup 5
done
echo -n "adagrad K=200 pairs:      "; one --opt adagrad --steps 200 --warmup 5
echo -n "adagrad K=200 base build: "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_base.so one --opt adagrad --steps 200 --warmup 5
echo -n "ucml128 censor adagrad K=20 pairs: "; one --model ucml --dim 128 --censor --opt adagrad --steps 20 --warmup 5
echo -n "ucml128 censor adagrad K=20 base:  "; ORX_LIB_PATH=$GRAFT_REPO_ROOT/scratch/lib_base.so one --model ucml --dim 128 --censor --opt adagrad --steps 20 --warmup 5
echo -n "sgd K=20 (unchanged kernel): "; one --steps 20 --warmup 5
timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pairwise.py tests/test_gpu_fullsize.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -6
