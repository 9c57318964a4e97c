#!/bin/bash
rm -rf gpurun_out/r4p; mkdir -p gpurun_out/r4p
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "3 0" "0 0"; do
  set -- $v
  ORX_GEMM16_DMA=$1 ORX_GEMM16_NTS=$2 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4p/prof -o p -- python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 20 --warmup 5 > gpurun_out/r4p/prof.log 2>&1
  f=$(find gpurun_out/r4p/prof -name '*kernel_trace.csv' | head -1)
  echo "== DMA=$1 NTS=$2"; python scripts/trace_last_step.py "$f" dlrm_loss_kernel > gpurun_out/r4p/step_dma$1_nts$2.txt; cat gpurun_out/r4p/step_dma$1_nts$2.txt
  rm -rf gpurun_out/r4p/prof
done
