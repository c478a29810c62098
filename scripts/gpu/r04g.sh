#!/bin/bash
# round 4: the whole GPU suite, then the bench exactly as the driver calls it (accuracy leg included)
out=/root/repo/gpurun_out/r04g
mkdir -p $out
cd /root/repo
timeout 2700 python -m pytest tests -m gpu -x -q > $out/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -4 $out/pytest_all.log
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$?"; tail -c 400 $out/bench_driver_args.err
python - <<PY
import json
d=json.load(open("$out/bench_driver_args.json"))
print("value", d["value"], d["timing"]["mode"], {k:v["value"] for k,v in d["timing"]["modes"].items()})
print("roofline", d["roofline"]["frac"], d["roofline"]["kernel_avg_us"])
print("accuracy", {k:d["rel_eigenvalue_error"].get(k) for k in ("value","max","train_seconds","train_steps_per_s","eval_seconds","not_measured_in_this_run","stepping")})
print("bf16x3", d.get("opt_in_path_bf16x3",{}).get("value"))
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
