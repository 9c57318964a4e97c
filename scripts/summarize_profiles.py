#!/usr/bin/env python3
"""Boils the rocprofv3 output of scripts/collect_profiles.sh down to the small files kept under profiles/."""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1]
out = "gpurun_out"


def find(d, suffix):
    hits = glob.glob(os.path.join(out, d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


stats = find(f"prof_{tag}", "kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats)))
    with open(f"{out}/{tag}_bench_kernel_stats.csv", "w", newline="") as f:
        csv.writer(f).writerows(rows)
trace = find(f"prof_{tag}", "kernel_trace.csv")
if trace:
    r = csv.DictReader(open(trace))
    keep = [row for row in r if "orx" in row["Kernel_Name"] or "fused" in row["Kernel_Name"] or "dedup" in row["Kernel_Name"]
            or "urgent" in row["Kernel_Name"] or "dup_apply" in row["Kernel_Name"]][:40]
    if keep:
        with open(f"{out}/{tag}_bench_kernel_trace_head.csv", "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys())); w.writeheader(); w.writerows(keep)

pmc = defaultdict(lambda: defaultdict(list))
for d, name in ((f"pmc_fetch_{tag}", "FETCH_SIZE"), (f"pmc_write_{tag}", "WRITE_SIZE")):
    p = find(d, "counter_collection.csv")
    if not p:
        continue
    for row in csv.DictReader(open(p)):
        if row["Counter_Name"] == name:
            pmc[row["Kernel_Name"]][name].append(float(row["Counter_Value"]))
if pmc:
    with open(f"{out}/{tag}_pmc_fused_summary.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "FETCH_SIZE_KiB_mean", "WRITE_SIZE_KiB_mean", "hbm_bytes_per_launch=(2*FETCH+WRITE)*1024"])
        for k, v in sorted(pmc.items()):
            fe, wr = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
            fm = sum(fe) / len(fe) if fe else float("nan"); wm = sum(wr) / len(wr) if wr else float("nan")
            w.writerow([k[:100], max(len(fe), len(wr)), f"{fm:.1f}", f"{wm:.1f}", f"{(2 * fm + wm) * 1024:.0f}"])
            if "fused_kernel<16" in k and fe and wr:
                json.dump({"bpr_d64_sgd": (2 * fm + wm) * 1024}, open(f"{out}/{tag}_traffic.json", "w"))
print(open(f"{out}/{tag}_bench.json").read() if os.path.exists(f"{out}/{tag}_bench.json") else "no bench line")
