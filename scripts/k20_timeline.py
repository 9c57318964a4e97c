"""usage: k20_timeline.py <kernel_trace.csv>: the kernels of the LAST train-step call in the trace (start offset, duration, name)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'plan_part_kernel<false>' in r['Kernel_Name']]
i0 = idx[-1] - 1 if idx else 0
t0 = int(rows[i0]['Start_Timestamp'])
fused = []
for r in rows[i0:]:
    st, en, n = int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']
    if 'fused_kernel' in n:
        fused.append((en - st) / 1e3)
        last = (en - t0) / 1e3
        continue
    if fused:
        print('   ... %d fused launches, mean %.2f us, ending at %.1f' % (len(fused), sum(fused) / len(fused), last)); fused = []
    print('%9.1f %7.1f  %s' % ((st - t0) / 1e3, (en - st) / 1e3, n[:60]))
