#!/bin/bash
# round 4: kernel stats of cfg2, cfg3 at its global batch (B = 4096) and cfg4 (chain kernel's cooperative row dot)
out=/root/repo/gpurun_out/r04h
mkdir -p $out
cd /root/repo
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_kernel_apply_gpu.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $out/pytest.log
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
run cfg3_b4096 --config cfg3 --batch-size 4096 --steps 100 --warmup 10 --repeats 3
run cfg3 --config cfg3 --steps 300 --warmup 20 --repeats 3
run cfg4 --config cfg4 --steps 100 --warmup 10 --repeats 3
