#!/bin/bash
# gemm16: static priority for waves 4-7 on (default) / off (NSVD_G16_DBG=16), cfg5 mixed kernel stats
cd /root/repo
for d in 0 16; do
  NSVD_G16_DBG=$d NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh r05s_dbg$d --config cfg5 --amp > /dev/null 2>&1
  python - <<PY
import csv, json
for r in csv.DictReader(open("gpurun_out/r05s_dbg$d/stats_kernel_stats.csv")):
    if "gemm16" in r["Name"]: print("dbg $d", r["Name"][-40:], round(float(r["AverageNs"]) / 1e3, 2))
print("dbg $d value", json.load(open("gpurun_out/r05s_dbg$d/bench.json"))["value"])
PY
done
