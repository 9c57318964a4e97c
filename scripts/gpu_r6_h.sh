#!/bin/bash
# round 6, call h: pairing in the pointwise steps (GMF / WRMF), GMF grid cap -- tests, A/B on one box
set -u
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pointwise.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_stepqueue.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
run() { # name model env...
  local n=$1 mdl=$2; shift 2
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model $mdl --steps 200 --warmup 20 > $O/${n}.json 2> $O/${n}.err
  python - <<P
import json
d=json.load(open("$O/${n}.json")); r=d["roofline"]; print("$n", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(r["kernel_us"],2), "frac", round(r["frac"],3), {k: round(v,1) for k,v in r["other_kernels_us"].items()})
P
}
for i in 1 2; do
run wrmf_pair_$i wrmf X=1
run wrmf_nopair_$i wrmf ORX_POINT_NO_PAIR=1
run gmf_pair_$i gmf X=1
run gmf_nopair_$i gmf ORX_POINT_NO_PAIR=1
run gmf_pair_g1024_$i gmf ORX_POINT_GRID_MAX=1024
run gmf_pair_g2048_$i gmf ORX_POINT_GRID_MAX=2048
run gmf_nopair_g1024_$i gmf ORX_POINT_NO_PAIR=1 ORX_POINT_GRID_MAX=1024
done
