"""usage: trace_by_shape.py <kernel_trace.csv> [min_us]: average duration per (kernel, grid, workgroup) of a rocprofv3 kernel trace"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.OrderedDict()
for r in rows:
    key = (r['Kernel_Name'][:64], r.get('Grid_Size', r.get('Grid_Size_X', '')), r.get('Workgroup_Size', r.get('Workgroup_Size_X', '')), r.get('LDS_Block_Size', ''))
    a = acc.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in acc.values())
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if v[1] / tot < 0.004: continue
    print('%-64s grid %-9s wg %-5s lds %-7s n %-5d avg %8.2f us  %5.1f %%' % (k[0], k[1], k[2], k[3], v[0], v[1] / v[0], 100 * v[1] / tot))
