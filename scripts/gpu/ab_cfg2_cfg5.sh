#!/bin/bash
# kernel stats of the headline and of cfg5 (mixed, float32) in one call:  ab_cfg2_cfg5.sh <tag>
tag=$1
cd /root/repo
for c in "cfg2:" "cfg5_amp:--config cfg5 --amp" "cfg5:--config cfg5"; do
  n=${c%%:*}; a=${c#*:}
  NSVD_PROFILE_PMC=0 timeout 900 bash scripts/collect_profiles.sh ${tag}_$n $a --accuracy off > /dev/null 2>&1
  python - <<PY
import csv, json
rows = list(csv.DictReader(open("gpurun_out/${tag}_$n/stats_kernel_stats.csv")))
print("== $n")
for r in rows[:9]:
    print("%-90s calls %6d avg %8.2f us" % (r["Name"][:90], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
d = json.load(open("gpurun_out/${tag}_$n/bench.json")); print("value", d["value"], "ms", d["ms_per_step"])
PY
done
