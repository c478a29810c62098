#!/bin/bash
# round 3, GPU call H: FusedKernelTrainer parity + cfg4 bench / kernel stats
cd /root/repo
timeout 900 python -m pytest tests/test_kernel_apply_gpu.py -m gpu -x -q 2>&1 | tail -4
NSVD_PROFILE_PMC=0 bash scripts/collect_profiles.sh r03h_cfg4 --config cfg4 > /dev/null 2>&1
head -c 1200 gpurun_out/r03h_cfg4/bench.json; echo; tail -3 gpurun_out/r03h_cfg4/bench.err
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/r03h_cfg4/stats_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={r['Percentage']}")
PY
