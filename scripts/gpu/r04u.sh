#!/bin/bash
# round 4: per-rank compute of the head-parallel sharding at world 8 / 4 / 2 (emulated on one GPU: L / N heads on 512 N rows)
out=/root/repo/gpurun_out/r04u
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for N in 8 4 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o hp$N -- python3 /root/repo/scripts/dev/hp_rank_prof.py $N > $out/hp$N.log 2>&1
  rm -f $out/hp${N}_kernel_trace.csv
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/hp${N}_kernel_stats.csv")))
print("== world $N")
tot=0
for r in rows[:8]:
    if int(r['Calls'])>=300:
        tot+=float(r['AverageNs'])/1e3*int(r['Calls'])/300
    print(f"  {r['Name'][:70]:<72}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
print("  kernel us per step ~", round(tot,1))
PY
done
