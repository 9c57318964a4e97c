#!/bin/bash
# round 6, call i: pointwise pairing A/B (forced on / auto / off) with the separate PAIR instantiation, workgroup-count caps
set -u
O=gpurun_out/r6i; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pointwise.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { # name model env...
  local n=$1 mdl=$2; shift 2
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model $mdl --steps 200 --warmup 20 > $O/${n}.json 2> $O/${n}.err
  python - <<P
import json
d=json.load(open("$O/${n}.json")); r=d["roofline"]; print("$n", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(r["kernel_us"],2), "frac", round(r["frac"],3), {k: round(v,1) for k,v in r["other_kernels_us"].items()})
P
}
for i in 1 2; do
for m in wrmf gmf; do
run ${m}_auto_$i $m X=1
run ${m}_forced_$i $m ORX_PAIR_ALWAYS=1
run ${m}_off_$i $m ORX_POINT_NO_PAIR=1
run ${m}_off_g8192_$i $m ORX_POINT_NO_PAIR=1 ORX_POINT_GRID_MAX=8192
run ${m}_off_g4096_$i $m ORX_POINT_NO_PAIR=1 ORX_POINT_GRID_MAX=4096
run ${m}_off_g2048_$i $m ORX_POINT_NO_PAIR=1 ORX_POINT_GRID_MAX=2048
done
done
