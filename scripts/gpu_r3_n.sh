#!/bin/bash
# round 3, call n: kernel timeline of the driver's protocol (one 20-step call)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/r3n
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/r3n/bench.json 2> /tmp/err.log
f=$(find /tmp/prof_tl -name '*kernel_trace.csv' | head -n 1)
cp "$f" $R/gpurun_out/r3n/kernel_trace.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed call: the last run of >= 20 fused_kernel launches before the profiling re-run; print the 2nd-to-last block of ~60 kernels around fused launches
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void fused_kernel")]
print(len(rows), "kernels;", len(idx), "fused launches")
# group fused launches into calls by gaps > 200 us
calls, cur = [], [idx[0]]
for a, b in zip(idx, idx[1:]):
    if int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"]) > 150000: calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
print([len(c) for c in calls])
for c in calls:
    if len(c) != 20: continue
    lo = c[0]
    while lo > 0 and int(rows[lo]["Start_Timestamp"]) - int(rows[lo - 1]["End_Timestamp"]) < 100000: lo -= 1
    hi = c[-1]
    while hi + 1 < len(rows) and int(rows[hi + 1]["Start_Timestamp"]) - int(rows[hi]["End_Timestamp"]) < 100000: hi += 1
    t0 = int(rows[lo]["Start_Timestamp"])
    print("---- call: %d kernels, %.1f us from first start to last end" % (hi - lo + 1, (int(rows[hi]["End_Timestamp"]) - t0) / 1e3))
    for r in rows[lo:hi + 1]:
        print("%8.1f %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:70]))
    break
PY
