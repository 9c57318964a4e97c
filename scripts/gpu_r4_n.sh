#!/bin/bash
mkdir -p gpurun_out/r4n
ORX_PAIR_ALWAYS=1 FUZZ_OPTS=sgd timeout 600 python scratch/fuzz_pairwise.py 300 7 > gpurun_out/r4n/fuzz_sgd_pair_always.log 2>&1; echo "rc=$?" >> gpurun_out/r4n/fuzz_sgd_pair_always.log
tail -4 gpurun_out/r4n/fuzz_sgd_pair_always.log
timeout 400 python scratch/fuzz_pairwise.py 150 11 > gpurun_out/r4n/fuzz_all.log 2>&1; echo "rc=$?" >> gpurun_out/r4n/fuzz_all.log
tail -3 gpurun_out/r4n/fuzz_all.log
