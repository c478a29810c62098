#!/bin/bash
# round 3, GPU call E: nsvd_cdk_step parity + cfg5 bench / kernel stats
out=/root/repo/gpurun_out/r03e
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_cdk_step_gpu.py tests/test_tower_gpu.py tests/test_cdk_gpu.py -m gpu -x -q 2>&1 | tail -25
NSVD_PROFILE_PMC=0 bash scripts/collect_profiles.sh r03e_cfg5 --config cfg5 > $out/collect.log 2>&1
cat gpurun_out/r03e_cfg5/bench.json | head -c 1500; echo
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r03e_cfg5/stats_kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
