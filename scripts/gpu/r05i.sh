#!/bin/bash
# round 5: the driver-style default bench line (with other_configs) and the bench launch tests
tag=${1:-r05i}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
SECONDS=0; timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$?"
echo "elapsed ${SECONDS}s"
python - <<PY
import json
d=json.load(open("$out/bench_driver_args.json"))
print("value", d["value"], d["timing"]["mode"], "roof", d["roofline"]["frac"], d["roofline"]["kernel_avg_us"])
print("acc", {k: d.get("rel_eigenvalue_error", {}).get(k) for k in ("value","max","train_seconds","eval_seconds","error")})
print("agree", d.get("opt_in_path_bf16x3", {}).get("agreement_of_the_hip_paths"))
for k, v in d.get("other_configs", {}).items():
    print(" ", k[:60], v.get("value"), v.get("ms_per_step"), (v.get("roofline") or {}).get("frac"), v.get("error"), v.get("measure_seconds"))
print("cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
print("not_measured" in json.dumps(d))
PY

