#!/bin/bash
# cfg5 mixed precision after the tower rebuild: kernel stats + bench line
tag=${1:-r05c}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_cfg5_amp --config cfg5 --amp > $out/collect_cfg5_amp.log 2>&1; echo "collect cfg5 amp rc=$?"
python - <<PY
import json, csv
d=json.load(open("$out/../${tag}_cfg5_amp/bench.json")); print("cfg5_amp", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_avg_us"], d["roofline"]["frac"])
rows = list(csv.DictReader(open("$out/../${tag}_cfg5_amp/stats_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = None
for r in rows[:22]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:8.2f} us {float(r['Percentage']):5.1f}%")
PY
