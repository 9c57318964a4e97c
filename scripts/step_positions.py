#!/usr/bin/env python3
"""Median duration of every launch of a DLRM step by its POSITION in the step (kernels of one name serve several layers):
scripts/step_positions.py <kernel_trace.csv> [marker-kernel-prefix]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2] if len(sys.argv) > 2 else 'head_bwd_kernel'
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith(marker)]
steps = [rows[idx[k]:idx[k + 1]] for k in range(10, len(idx) - 1)]
n = collections.Counter(len(s) for s in steps).most_common(1)[0][0]
steps = [s for s in steps if len(s) == n]
tot = 0.0
for p in range(n):
    d = sorted((int(s[p]['End_Timestamp']) - int(s[p]['Start_Timestamp'])) / 1e3 for s in steps)
    tot += d[len(d) // 2]
    print('%2d %-58s grid %-8s median %6.2f us  min %6.2f' % (p, steps[0][p]['Kernel_Name'][:58], steps[0][p]['Grid_Size_X'], d[len(d) // 2], d[0]))
print('launches %d, sum of medians %.1f us over %d steps' % (n, tot, len(steps)))
