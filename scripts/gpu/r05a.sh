#!/bin/bash
# round 5, baseline of the tree as round 4 left it: cfg5 (float32 and mixed precision) + cfg4 kernel stats and bench lines
tag=${1:-r05a}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
for cfg in cfg5 cfg4; do
  NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_$cfg --config $cfg --accuracy off > $out/collect_$cfg.log 2>&1; echo "collect $cfg rc=$?"
done
NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_cfg5_amp --config cfg5 --amp --accuracy off > $out/collect_cfg5_amp.log 2>&1; echo "collect cfg5 amp rc=$?"
for d in cfg5 cfg4 cfg5_amp; do python - <<PY
import json
try:
    d=json.load(open("$out/../${tag}_$d/bench.json")); print("$d", d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_avg_us"], d["roofline"]["frac"])
except Exception as e: print("$d", "failed", e)
PY
done
