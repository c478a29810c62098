#!/bin/bash
# round 6: the float16 half type + GradScaler of the CDK step: parity vs the oracle, then timing
out=/root/repo/gpurun_out/r06d
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests/test_tower_gpu.py tests/test_cdk_step_gpu.py tests/test_gemm16_gpu.py tests/test_cdk_gpu.py -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log
for dt in bfloat16 float16; do
  timeout 300 python bench.py --config cfg5 --amp --amp-dtype $dt --no-cpu-baseline > $out/bench_cfg5_$dt.json 2> $out/bench_cfg5_$dt.err
  python - <<PY
import json
d = json.load(open("$out/bench_cfg5_$dt.json")); r = d["roofline"]
print("$dt", d["value"], d["ms_per_step"], r["kernel"], r["kernel_avg_us"], d.get("grad_scaler"), d["final_loss"])
PY
done
