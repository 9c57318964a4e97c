#!/bin/bash
# MFMA / LDS counters of the DLRM fp16 step, per kernel: scripts/pmc_dlrm.sh <tag>  ->  gpurun_out/<tag>_dlrm_pmc.csv
# (counter passes only: --pmc with --kernel-trace, never with other trace domains)
TAG=${1:-p}
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --no-cpu-baseline --model dlrm --fp16-mlp --steps 10 --warmup 3"
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmcd_${TAG}_$N -o c -- $CMD > $OUT/${TAG}_pmc_$N.log 2>&1 || echo "pass $N failed"
done
cd $REPO
python - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/pmcd_${TAG}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in acc.values() for c in k})
with open("$OUT/${TAG}_dlrm_pmc.csv", "w", newline="") as fo:
    w = csv.writer(fo); w.writerow(["kernel", "dispatches"] + [n + "_mean" for n in names])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        if not any(x in k for x in ("gemm16", "slab_reduce", "head_", "interact", "tiny", "apply_rows", "cast16", "dense_shadow")): continue
        w.writerow([k, max(len(x) for x in v.values())] + [f"{sum(v[n]) / len(v[n]):.0f}" if n in v else "" for n in names])
print(open("$OUT/${TAG}_dlrm_pmc.csv").read())
PY
