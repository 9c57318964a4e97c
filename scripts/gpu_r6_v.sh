#!/bin/bash
# experiment (ORX_EXP_TWICE build): which buffer's state costs the top MLP's first product 6.6 us behind the interaction forward?  A small kernel reads one
# byte per 4 KB (translations only; +16: one byte per 128 B, the data too) of: 1 = its input R16, 2 = the weights, 4 = its output, 8 = the relu mask words
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6v; mkdir -p $O
for v in 64 192; do
cd /tmp
ORX_GEMM16_WARM=0 ORX_EXP_TOP0_TOUCH=$v rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -o g -- python $GRAFT_REPO_ROOT/bench.py --model dlrm --fp16-mlp --no-cpu-baseline --steps 30 --warmup 10 > $GRAFT_REPO_ROOT/$O/prof_$v.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_$v -name "*kernel_trace.csv" | head -1); echo "== touch $v"; python scripts/step_positions.py $f | sed -n 19,22p
done
