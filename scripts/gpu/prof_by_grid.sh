#!/bin/bash
# rocprofv3 --kernel-trace of a python script; per (kernel, grid) average durations - tells the launches of one kernel apart
#   prof_by_grid.sh <tag> <script.py> [args]
tag=$1; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o stats -- python3 "$@" > $out/run.log 2>&1
python3 - <<PY
import csv, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open("$out/stats_kernel_trace.csv")):
    key = (r['Kernel_Name'][:70], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r.get('Workgroup_Size_X'))
    a = acc.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(a[1] for a in acc.values())
for k, a in sorted(acc.items(), key=lambda kv: -kv[1][1])[:${TOPN:-30}]:
    print(f"{k[0]:70s} grid {k[1]:>7s} {k[2]:>5s} {k[3]:>4s} calls {a[0]:5d} avg {a[1]/a[0]:9.2f} us  {100*a[1]/tot:5.1f}%")
PY
rm -f $out/stats_kernel_trace.csv
tail -${TAILN:-6} $out/run.log
