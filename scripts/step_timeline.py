#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace as a timeline: scripts/step_timeline.py gpurun_out/prof_<tag>/b_kernel_trace.csv [marker-kernel]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else 'dense_shadow_kernel'
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(marker)]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>4s}  {r['Kernel_Name'][:64]}")
print(f"step total {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
