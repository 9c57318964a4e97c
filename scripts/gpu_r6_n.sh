#!/bin/bash
# round 6, call n: the two plan levers again, 10 interleaved repetitions of the driver's protocol (K = 20) on one box; tn two-per-CU form in the DLRM step
set -u
O=gpurun_out/r6n; mkdir -p $O
for i in 1 2 3 4 5 6 7 8 9 10; do
for v in base p1batch noswap both; do
  case $v in base) E="X=1";; p1batch) E="ORX_PLAN_P1_BATCH=1";; noswap) E="ORX_PLAN_NO_SWAP=1";; both) E="ORX_PLAN_P1_BATCH=1 ORX_PLAN_NO_SWAP=1";; esac
  env $E timeout 300 python bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $O/k20_${v}_$i.json 2> /dev/null
done
done
python - <<'P'
import json, glob, statistics as st
for v in ("base", "p1batch", "noswap", "both"):
    xs = sorted(json.load(open(f))["ms_per_step"] * 1000 for f in glob.glob(f"gpurun_out/r6n/k20_{v}_*.json"))
    ks = sorted(json.load(open(f))["roofline"]["kernel_us"] for f in glob.glob(f"gpurun_out/r6n/k20_{v}_*.json"))
    print(f"{v:8s} n={len(xs)} step us: median {st.median(xs):.2f} mean {st.mean(xs):.2f} min {xs[0]:.2f} max {xs[-1]:.2f} | fused kernel median {st.median(ks):.2f}")
P
run() { # name env...
  local n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 40 --warmup 10 > $O/dlrm_$n.json 2> $O/dlrm_$n.err
  python - <<P
import json
d=json.load(open("$O/dlrm_$n.json")); print("$n", round(d["ms_per_step"]*1000,1), "us/step  products", round(d["roofline"]["gemm_ms_per_step"]*1000,1))
P
}
run base X=1
run tn2 ORX_GEMM16_TN_DMA=2 ORX_GEMM16_TN_PER_CU=2
run base2 X=1
run tn2b ORX_GEMM16_TN_DMA=2 ORX_GEMM16_TN_PER_CU=2
