#!/bin/bash
# kernel stats + bench line of one configuration:  stats_cfg.sh <tag> <bench.py arguments ...>
tag=$1; shift
cd /root/repo
NSVD_PROFILE_PMC=0 timeout 900 bash scripts/collect_profiles.sh $tag "$@" > gpurun_out/${tag}_collect.log 2>&1; echo "collect rc=$?"
python - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/$tag/stats_kernel_stats.csv")))
for r in rows[:14]:
    print("%-100s calls %6d avg %8.2f us %5.1f%%" % (r["Name"][:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
d = json.load(open("gpurun_out/$tag/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel us", d["roofline"].get("kernel_avg_us"))
PY
