#!/bin/bash
# round 6, call m: the two plan levers DESIGN section 8 listed as unmeasured -- A/B on one box (K = 20 driver protocol and K = 200), pairing tests under both
set -u
O=gpurun_out/r6m; mkdir -p $O
for e in "X=1" "ORX_PLAN_P1_BATCH=1" "ORX_PLAN_NO_SWAP=1"; do
  env $e timeout 900 python -m pytest tests/test_gpu_pairing.py tests/test_gpu_pairwise.py -q -m gpu -x > $O/tests_$(echo $e | tr '=' '_').log 2>&1; echo "tests [$e] rc=$?"; tail -1 $O/tests_$(echo $e | tr '=' '_').log
done
run() { # name steps warm env...
  local n=$1 st=$2 wu=$3; shift 3
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps $st --warmup $wu > $O/${n}.json 2> $O/${n}.err
  python - <<P
import json
d=json.load(open("$O/${n}.json")); r=d["roofline"]; print("$n", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(r["kernel_us"],2), {k: round(v,1) for k,v in r["other_kernels_us"].items()})
P
}
for i in 1 2 3; do
run k20_base_$i 20 5 X=1
run k20_p1batch_$i 20 5 ORX_PLAN_P1_BATCH=1
run k20_noswap_$i 20 5 ORX_PLAN_NO_SWAP=1
run k20_both_$i 20 5 ORX_PLAN_P1_BATCH=1 ORX_PLAN_NO_SWAP=1
done
for i in 1 2; do
run k200_base_$i 200 20 X=1
run k200_p1batch_$i 200 20 ORX_PLAN_P1_BATCH=1
run k200_noswap_$i 200 20 ORX_PLAN_NO_SWAP=1
done
