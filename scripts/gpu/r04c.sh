#!/bin/bash
# round 4: bf16x3 layer 0 v2 (W fragments straight to registers, deep prefetch): accuracy + kernel times
out=/root/repo/gpurun_out/r04c
mkdir -p $out
cd /root/repo
timeout 300 python scripts/dev/bf3_check.py > $out/bf3_check.log 2>&1; echo "bf3_check rc=$?"; cat $out/bf3_check.log
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -k "bf16x3" > $out/pytest_bf3.log 2>&1; echo "pytest bf3 rc=$?"; tail -3 $out/pytest_bf3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bf3 -- python3 /root/repo/bench.py --path bf16x3 --steps 300 --warmup 20 --repeats 3 --no-cpu-baseline --no-extras --accuracy off --graph off > $out/bench_bf3.json 2> $out/bench_bf3.err
rm -f $out/bf3_kernel_trace.csv
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/bf3_kernel_stats.csv")))
for r in rows[:6]:
    print(f"{r['Name'][:70]:<72}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
d = json.load(open("$out/bench_bf3.json")); print("steps/s", d["value"], "ms/step", d["ms_per_step"], "loss", d["final_loss"], d["params_finite"])
PY
