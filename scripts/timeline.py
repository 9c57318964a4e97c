#!/usr/bin/env python3
"""Timeline of the LAST K-step call in a rocprofv3 --kernel-trace CSV (start offset, duration, gap to the previous kernel).
usage: scripts/timeline.py <kernel_trace.csv> [marker substring = plan_part_kernel<false>] [occurrence from the end = 2]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else 'plan_part_kernel<false>'
occ = int(sys.argv[3]) if len(sys.argv) > 3 else 2
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
i0 = max(idx[-occ] - 4, 0)
i1 = idx[-occ + 1] if occ > 1 else len(rows)
t0 = int(rows[i0]['Start_Timestamp']); prev = None
for r in rows[i0:i1]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    gap = '' if prev is None else f"gap {(s - prev) / 1e3:7.1f}"
    name = r['Kernel_Name'][:56]
    if 'fused_kernel' in name and prev is not None and abs(s - prev) < 200 and 'fused' in last: 
        nf += 1; prev = e; continue
    print(f"{s / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap:12s} {name}")
    prev = e; last = name; nf = 0
