#!/bin/bash
# quick kernel stats at cfg2 (native and bf16x3)
out=/root/repo/gpurun_out/r04j
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 /root/repo/bench.py "$@" --no-cpu-baseline --no-extras --accuracy off --graph off > $out/bench_$name.json 2> $out/bench_$name.err
  rm -f $out/${name}_kernel_trace.csv
  python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${name}_kernel_stats.csv")))
print("== $name")
for r in rows[:4]:
    print(f"  {r['Name'][:66]:<68}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
try:
    d = json.load(open("$out/bench_$name.json")); print("  steps/s", d["value"], "ms/step", d["ms_per_step"])
except Exception as e: print("  bench:", e, open("$out/bench_$name.err").read()[-800:])
PY
}
run cfg2 --steps 300 --warmup 20 --repeats 3
run cfg2_bf16x3 --path bf16x3 --steps 300 --warmup 20 --repeats 3
if [ -n "$OLD_LIB" ]; then export NSVD_LIB_PATH=$OLD_LIB; run cfg2_old --steps 300 --warmup 20 --repeats 3; run cfg2_bf16x3_old --path bf16x3 --steps 300 --warmup 20 --repeats 3; fi
