#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pairwise.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_stress.py tests/test_gpu_stepqueue.py tests/test_gpu_c4_shapes.py -x -q -m gpu 2>&1 | tail -6
one() { timeout 200 python bench.py --no-cpu-baseline --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_us'],2), 'frac', round(d['roofline']['frac'],3), d['roofline'].get('other_kernels_us'))"; }
for i in 1 2 3; do echo -n "sgd K=20: "; one --steps 20 --warmup 5; done
echo -n "sgd K=200: "; one --steps 200 --warmup 5
echo -n "ucml128 censor K=20: "; one --model ucml --dim 128 --censor --steps 20 --warmup 5
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5_k20b -o p -- python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > /dev/null 2>&1
f=$(find gpurun_out/r5_k20b -name '*kernel_trace.csv' | head -1)
python scripts/k20_timeline.py "$f"
rm -rf gpurun_out/r5_k20b
