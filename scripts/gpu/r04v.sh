#!/bin/bash
# round 4: kernel breakdown of the generic path (forward + loss + backward at configs[1]'s sizes, hidden width 128 forced generic, 64, 256, 96)
out=/root/repo/gpurun_out/r04v
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o gen -- python3 /root/repo/scripts/dev/generic_time.py > $out/gen.log 2>&1
rm -f $out/gen_kernel_trace.csv
tail -4 $out/gen.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/gen_kernel_stats.csv")))
for r in rows[:12]:
    print(f"  {r['Name'][:80]:<82}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us {float(r['Percentage']):>6.1f}%")
PY
