#!/bin/bash
# round 5, call n: kernel trace of the row-sharded engine on one rank (which launches make up the 82 us step)
set -u
O=gpurun_out/r5n; mkdir -p $O
R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o b -- python $R/bench.py --no-cpu-baseline --sharded --steps 128 --warmup 64 > $R/$O/prof.log 2>&1
cd $R
f=$(ls $O/prof/*/b_kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$O/prof/b_kernel_stats.csv
python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), ("%.2f"%(float(r['AverageNs'])/1000)).rjust(8), r['Percentage'])
P
t=$(ls $O/prof/*/b_kernel_trace.csv 2>/dev/null | head -1); [ -z "$t" ] && t=$O/prof/b_kernel_trace.csv
python - "$t" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 40 launches
t0=int(rows[-60]['Start_Timestamp'])
for r in rows[-60:-20]:
    print("%9.1f %7.2f  %s"%((int(r['Start_Timestamp'])-t0)/1000,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000,r['Kernel_Name'][:100]))
P
