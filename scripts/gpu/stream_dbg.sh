#!/bin/bash
# streaming backward, diagnostic variants: kernel time with NSVD_STREAM_DBG = 0 / 1 / 2 / 3
cd /root/repo
for d in 0 1 2 3; do
  NSVD_STREAM_DBG=$d NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh r05q_dbg$d --config cfg4 > /dev/null 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/r05q_dbg$d/stats_kernel_stats.csv")):
    if "stream_bwd" in r["Name"]: print("dbg $d", round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
