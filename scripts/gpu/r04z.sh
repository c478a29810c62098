#!/bin/bash
# round 4, evidence for profiles/: full GPU suite, kernel stats + counters (native and bf16x3), the other configs'
# kernel stats, driver-style bench lines (1 GPU; 2 ranks over gloo on one device; RCCL in a world of one)
tag=${1:-r04z}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
timeout 2700 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 900 bash scripts/collect_profiles.sh ${tag}_cfg2 --accuracy off > $out/collect_cfg2.log 2>&1; echo "collect cfg2 rc=$?"
timeout 900 bash scripts/collect_profiles.sh ${tag}_cfg2_bf16x3 --path bf16x3 --accuracy off > $out/collect_cfg2_bf16x3.log 2>&1; echo "collect cfg2 bf16x3 rc=$?"
for cfg in cfg1 cfg3 cfg4; do
  NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_$cfg --config $cfg --accuracy off > $out/collect_$cfg.log 2>&1; echo "collect $cfg rc=$?"
done
NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_cfg3_b4096 --config cfg3 --batch-size 4096 --accuracy off > $out/collect_cfg3_b4096.log 2>&1; echo "collect cfg3 b4096 rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$?"
NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --accuracy off > $out/bench_n2_gloo.json 2> $out/bench_n2_gloo.err; echo "bench n2 rc=$?"
timeout 600 python bench.py --gpus 1 --force-exchange --steps 200 --warmup 20 --accuracy off > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err; echo "bench rccl1 rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_driver_args.json"))
print("value", d["value"], d["timing"]["mode"], {k:v["value"] for k,v in d["timing"]["modes"].items()})
print("roofline", d["roofline"]["frac"], d["roofline"]["kernel_avg_us"], d["roofline"].get("traffic"))
print("accuracy", {k:d["rel_eigenvalue_error"].get(k) for k in ("value","max","train_seconds","train_steps_per_s","eval_seconds","not_measured_in_this_run")})
print("parity", d["eigenvalue_parity_vs_float64_oracle"]["default_mode_laplacian_eps_0.01"]["hip_fp32_vs_f64_oracle"])
print("bf16x3", d.get("opt_in_path_bf16x3",{}).get("value"))
print("cpu", d["cpu_baseline"]["value"], d["speedup_vs_cpu_baseline"])
PY
