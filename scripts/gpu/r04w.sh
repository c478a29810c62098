#!/bin/bash
# round 4: partial-moment kernels with independent accumulator chains - kernel stats at cfg4 and cfg3 (B = 4096)
out=/root/repo/gpurun_out/r04w
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 /root/repo/bench.py "$@" --no-cpu-baseline --no-extras --accuracy off --graph off > $out/bench_$name.json 2> $out/bench_$name.err
  rm -f $out/${name}_kernel_trace.csv
  python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${name}_kernel_stats.csv")))
print("== $name")
for r in rows[:9]:
    print(f"  {r['Name'][:66]:<68}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
try:
    d = json.load(open("$out/bench_$name.json")); print("  steps/s", d["value"], "ms/step", d["ms_per_step"])
except Exception as e: print("  bench:", e, open("$out/bench_$name.err").read()[-800:])
PY
}
run cfg4 --config cfg4 --steps 100 --warmup 10 --repeats 3
run cfg3_b4096 --config cfg3 --batch-size 4096 --steps 100 --warmup 10 --repeats 3
