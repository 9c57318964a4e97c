#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV: calls, mean / total duration, share.
usage: scripts/trace_summary.py <kernel_trace.csv> [top]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = r['Kernel_Name'].split('(')[0][:64]
    agg[k][0] += 1; agg[k][1] += d
tot = sum(v[1] for v in agg.values())
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:66s} calls {n:6d}  mean {t/n:9.2f} us  total {t/1e3:9.3f} ms  {100*t/tot:5.1f}%")
