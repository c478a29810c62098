#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python script; prints the top kernels (name truncated, calls, avg us)
#   prof_kernels.sh <tag> <script.py> [args]
tag=$1; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o stats -- python3 "$@" > $out/run.log 2>&1
rm -f $out/stats_kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/stats_kernel_stats.csv")))
for r in rows[:${TOPN:-24}]:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  {float(r['Percentage']):5.1f}%")
PY
tail -${TAILN:-8} $out/run.log
