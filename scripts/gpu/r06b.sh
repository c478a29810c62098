#!/bin/bash
# round 6: the whole-column BatchNorm-in-epilogue kernels (csrc/tower_col.h): parity, then timing against the strips
out=/root/repo/gpurun_out/r06b
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests/test_tower_gpu.py tests/test_cdk_step_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log
timeout 300 python scripts/dev/tcol_time.py > $out/tcol_time_a.log 2>&1; cat $out/tcol_time_a.log
NSVD_TCOL_FORM=b timeout 300 python scripts/dev/tcol_time.py > $out/tcol_time_b.log 2>&1; cat $out/tcol_time_b.log
for f in 1 0; do
  NSVD_TOWER16_FUSED=$f timeout 300 python bench.py --config cfg5 --amp --no-cpu-baseline > $out/bench_cfg5_amp_fused$f.json 2> $out/bench_cfg5_amp_fused$f.err
  python - <<PY
import json
d = json.load(open("$out/bench_cfg5_amp_fused$f.json")); r = d["roofline"]
print("fused=$f", d["value"], d["ms_per_step"], r["kernel"], r["kernel_avg_us"], r["bound"], r["frac"], r.get("step"))
PY
done
NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh r06b_cfg5_amp --config cfg5 --amp > $out/collect.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/r06b_cfg5_amp/stats_kernel_stats.csv")))
for r in rows[:24]:
    print("%-100s calls %6d avg %8.2f us %5.1f%%" % (r["Name"][:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
