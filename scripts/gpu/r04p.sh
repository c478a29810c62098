#!/bin/bash
# round 4: persistent chain kernel - bit identity against the block-per-workgroup form, cfg3 at B = 4096 and cfg4 A/B
out=/root/repo/gpurun_out/r04p
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_kernel_apply_gpu.py tests/test_dropin_gpu.py -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $name -- python3 /root/repo/bench.py "$@" --no-cpu-baseline --no-extras --accuracy off --graph off > $out/bench_$name.json 2> $out/bench_$name.err
  rm -f $out/${name}_kernel_trace.csv
  python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$out/${name}_kernel_stats.csv")))
print("== $name")
for r in rows[:5]:
    print(f"  {r['Name'][:66]:<68}{int(r['Calls']):>7}{float(r['AverageNs'])/1e3:>10.2f} us")
try:
    d = json.load(open("$out/bench_$name.json")); print("  steps/s", d["value"], "ms/step", d["ms_per_step"])
except Exception as e: print("  bench:", e, open("$out/bench_$name.err").read()[-800:])
PY
}
run cfg3_b4096 --config cfg3 --batch-size 4096 --steps 100 --warmup 10 --repeats 3
run cfg4 --config cfg4 --steps 100 --warmup 10 --repeats 3
export NSVD_CHAIN_PERSIST=0
run cfg3_b4096_block --config cfg3 --batch-size 4096 --steps 100 --warmup 10 --repeats 3
run cfg4_block --config cfg4 --steps 100 --warmup 10 --repeats 3
