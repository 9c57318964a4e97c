#!/bin/bash
# experiment: the top MLP's first product reads its own CODE as data at its start (ORX_GEMM16_CODE_PF=bytes) -- is its 6 us behind the interaction forward
# the instruction fetch from a cold L2?  (the kernel is 68 KB of code)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6x; mkdir -p $O
for v in 0 32768 61440; do
cd /tmp
ORX_GEMM16_CODE_PF=$v rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 30 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); echo "== code prefetch $v"; python scripts/step_positions.py $f | sed -n 19,22p; tail -2 $O/prof_$v.log | cut -c1-200
done
