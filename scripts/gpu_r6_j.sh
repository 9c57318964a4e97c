#!/bin/bash
# round 6, call j: closed-form replay of the lazily applied TF-2.0 Adam -- tests, A/B on one box, kernel stats
set -u
O=gpurun_out/r6j; mkdir -p $O
REPO=$(pwd)
timeout 1800 python -m pytest tests/test_gpu_pairwise.py tests/test_gpu_fullsize.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_edges.py tests/test_gpu_refstub.py tests/test_gpu_stepqueue.py tests/test_gpu_reference_examples.py tests/test_gpu_compose.py tests/test_gpu_stress.py -q -m gpu -x -k "adam or Adam or golden or fuzz or example or stress or lazy" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --opt adam --steps 128 --warmup 64 > $O/${n}.json 2> $O/${n}.err
  python - <<P
import json
d=json.load(open("$O/${n}.json")); r=d["roofline"]; print("$n", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(r["kernel_us"],2), "frac", round(r["frac"],3), {k: round(v,1) for k,v in r["other_kernels_us"].items()})
P
}
run cf_1 X=1
run loops_1 ORX_ADAM_NO_CF=1
run cf_2 X=1
run loops_2 ORX_ADAM_NO_CF=1
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/prof -o b -- python $REPO/bench.py --no-cpu-baseline --opt adam --steps 128 --warmup 64 > $REPO/$O/prof.log 2>&1)
S=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $S $O/kernel_stats.csv; head -8 $S | cut -c1-160
