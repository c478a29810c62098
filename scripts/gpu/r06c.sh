#!/bin/bash
# round 6: (1) the strips form of the mixed tower after the rebuild; (2) the weight-gradient kernel's phase-shifted
# schedule (64-wide dW_0 tiles, two per CU, the second started late): headline steps/s per delay
out=/root/repo/gpurun_out/r06c
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_tower_gpu.py tests/test_cdk_step_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log
run() {
  python bench.py --accuracy off --no-extras --no-cpu-baseline --graph off --steps 500 --warmup 100 --repeats 7 > $out/b.json 2> $out/b.err
  python -c "
import json; d = json.load(open('$out/b.json')); print('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_avg_us'])"
}
run base
run base_again
for dl in 0 400 800 1200 1600 2000; do
  NSVD_WG_TW64=1 NSVD_WG_DELAY=$dl run tw64_delay$dl
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o st_base -- python3 /root/repo/bench.py --accuracy off --no-extras --no-cpu-baseline --graph off --steps 300 --repeats 3 > $out/prof_base.log 2>&1
NSVD_WG_TW64=1 NSVD_WG_DELAY=1200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o st_tw64 -- python3 /root/repo/bench.py --accuracy off --no-extras --no-cpu-baseline --graph off --steps 300 --repeats 3 > $out/prof_tw64.log 2>&1
rm -f $out/st_*_kernel_trace.csv
for tag in base tw64; do
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$out/st_${tag}_kernel_stats.csv")))
for r in rows[:4]:
    print("$tag  %-80s calls %6d avg %8.2f us" % (r["Name"][:80], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
done
