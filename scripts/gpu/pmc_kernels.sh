#!/bin/bash
# rocprofv3 --pmc <counters> of a python script (kernel-trace only), per-kernel averages
#   pmc_kernels.sh <tag> "<counters>" <script.py> [args]
tag=$1; ctr=$2; shift; shift
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o pmc -- python3 "$@" > $out/run.log 2>&1
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$out/pmc_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if len(next(iter(d.values()))) < 20: continue
    print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
rm -f $out/pmc_kernel_trace.csv $out/pmc_counter_collection.csv
