#!/bin/bash
# round 4: A/B on ONE box - bf16x3 K loop with one barrier per chunk (scripts/_diag/libnsvd_hip_bf3chunk.so) against one
# barrier per pair (the tree): forward-only loop under rocprofv3
out=/root/repo/gpurun_out/r04f
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in pair chunk pair chunk; do
  if [ $v = chunk ]; then export NSVD_LIB_PATH=/root/repo/scripts/_diag/libnsvd_hip_bf3chunk.so; else unset NSVD_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ab_$v -- python3 /root/repo/bench.py --path bf16x3 --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-extras --accuracy off --graph off > $out/bench_$v.json 2> $out/bench_$v.err
  rm -f $out/ab_${v}_kernel_trace.csv
  python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/ab_${v}_kernel_stats.csv")))
for r in rows[:1]:
    print("$v", f"{r['Name'][:60]:<62}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
d = json.load(open("$out/bench_$v.json")); print("   steps/s", d["value"])
PY
done
