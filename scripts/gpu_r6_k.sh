#!/bin/bash
# round 6, call k: closed-form Adam in the pointwise steps (tests, A/B), VALU counters of the pairwise Adam kernel (loops vs closed form),
# what a ncclSend / ncclRecv group costs on one rank
set -u
O=gpurun_out/r6k; mkdir -p $O
REPO=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_pointwise.py tests/test_gpu_api.py tests/test_gpu_fuzz.py tests/test_gpu_compose.py -q -m gpu -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
run() { # name model env...
  local n=$1 mdl=$2; shift 2
  env "$@" timeout 300 python bench.py --no-cpu-baseline --model $mdl --opt adam --steps 128 --warmup 64 > $O/${n}.json 2> $O/${n}.err
  python - <<P
import json
d=json.load(open("$O/${n}.json")); r=d["roofline"]; print("$n", round(d["ms_per_step"]*1000,2), "us/step  kernel", round(r["kernel_us"],2))
P
}
run wrmf_adam_cf wrmf X=1
run wrmf_adam_loops wrmf ORX_ADAM_NO_CF=1
run gmf_adam_cf gmf X=1
run gmf_adam_loops gmf ORX_ADAM_NO_CF=1
# VALU counters of the pairwise Adam step (separate --pmc passes with --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
for V in cf loops; do
  if [ $V = loops ]; then export ORX_ADAM_NO_CF=1; else unset ORX_ADAM_NO_CF; fi
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    N=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/$O/pmc_${V}_$N -o c -- python $REPO/bench.py --no-cpu-baseline --no-secondary --opt adam --steps 16 --warmup 48 > $REPO/$O/pmc_${V}_$N.log 2>&1 || echo "pass $V $N failed"
  done
done
unset ORX_ADAM_NO_CF
cd $REPO
python - <<'PY'
import csv, glob
from collections import defaultdict
for V in ("cf", "loops"):
    acc = defaultdict(list)
    for f in glob.glob(f"gpurun_out/r6k/pmc_{V}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void fused_kernel"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(V, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())}, "launches", {k: len(v) for k, v in acc.items()})
PY
ORX_SHARD_RCCL_SELF=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 scripts/rccl_group_latency.py > $O/rccl_group_latency.txt 2>&1; tail -8 $O/rccl_group_latency.txt
