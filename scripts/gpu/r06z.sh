#!/bin/bash
# round 6, evidence for profiles/: the full GPU suite; kernel stats + counters (headline; cfg5 mixed), the other configs'
# kernel stats; driver-style bench lines (1 GPU; 2 ranks over gloo on one device, dp and hp with uneven heads; RCCL in a
# world of one).   r06z.sh <tag> [tests]   - "tests" runs the full GPU suite only
tag=${1:-r06z}
out=/root/repo/gpurun_out/$tag
mkdir -p $out
cd /root/repo
if [ "$2" = "tests" ]; then
  SECONDS=0; timeout 3000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$? in ${SECONDS}s" >> $out/pytest.log
  tail -5 $out/pytest.log
  exit 0
fi
timeout 900 bash scripts/collect_profiles.sh ${tag}_cfg2 --accuracy off > $out/collect_cfg2.log 2>&1; echo "collect cfg2 rc=$?"
timeout 900 bash scripts/collect_profiles.sh ${tag}_cfg5_amp --config cfg5 --amp > $out/collect_cfg5_amp.log 2>&1; echo "collect cfg5 amp rc=$?"
for cfg in cfg1 cfg3 cfg4 cfg5; do
  NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_$cfg --config $cfg > $out/collect_$cfg.log 2>&1; echo "collect $cfg rc=$?"
done
NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh ${tag}_cfg5_f16 --config cfg5 --amp --amp-dtype float16 > $out/collect_cfg5_f16.log 2>&1; echo "collect cfg5 f16 rc=$?"
# the generic path (hidden widths 256 / 64 at configs[1]'s sizes): per-(kernel, grid) launch durations, SQ counters
TOPN=16 TAILN=2 timeout 300 scripts/gpu/prof_by_grid.sh ${tag}_gen256 /root/repo/scripts/dev/generic_time.py 256 > $out/generic_h256_by_grid.txt 2>&1
TOPN=16 TAILN=2 timeout 300 scripts/gpu/prof_by_grid.sh ${tag}_gen64 /root/repo/scripts/dev/generic_time.py 64 > $out/generic_h64_by_grid.txt 2>&1
timeout 300 scripts/gpu/pmc_kernels.sh ${tag}_gen256_pmc "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" /root/repo/scripts/dev/generic_time.py 256 > $out/generic_h256_pmc_sq.txt 2>&1
timeout 300 scripts/gpu/pmc_kernels.sh ${tag}_gen256_pmc2 "TCC_HIT_sum TCC_MISS_sum" /root/repo/scripts/dev/generic_time.py 256 > $out/generic_h256_pmc_l2.txt 2>&1
echo "generic evidence done"
SECONDS=0; timeout 1200 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$? in ${SECONDS}s"
NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --accuracy off > $out/bench_n2_gloo.json 2> $out/bench_n2_gloo.err; echo "bench n2 rc=$?"
timeout 600 python bench.py --gpus 1 --force-exchange --steps 200 --warmup 20 --accuracy off > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err; echo "bench rccl1 rc=$?"
python - <<PY
import json
d=json.load(open("$out/bench_driver_args.json"))
print("value", d["value"], d["timing"]["mode"], "roof", d["roofline"]["frac"], d["roofline"]["kernel_avg_us"])
print("acc", {k: d.get("rel_eigenvalue_error", {}).get(k) for k in ("value","max","train_seconds","eval_seconds","error")})
for k, v in d.get("other_configs", {}).items():
    r = v.get("roofline") or {}
    print(" ", k[:70], v.get("value"), v.get("ms_per_step"), r.get("bound"), r.get("frac"), r.get("step_frac"), v.get("error"))
print("cpu", d["cpu_baseline"]["value"] if d.get("cpu_baseline") else None)
for f in ("bench_n2_gloo", "bench_rccl_world1"):
    try:
        e = json.load(open("$out/%s.json" % f)); print(f, e["value"], e.get("metric", "")[:80])
    except Exception as ex:
        print(f, "unreadable", ex)
PY
