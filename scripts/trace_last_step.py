"""usage: trace_last_step.py <kernel_trace.csv> <marker substring> : the launches from the last-but-one occurrence of the marker kernel to the last one (one step), in order"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
lo, hi = idx[-2], idx[-1]
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print('%9.1f %7.2f  grid %-8s wg %-4s %s' % ((st - t0) / 1e3, (en - st) / 1e3, r.get('Grid_Size', ''), r.get('Workgroup_Size', ''), r['Kernel_Name'][:90]))
