#!/bin/bash
# round 3, final tree: full GPU suite, evidence for profiles/ (kernel stats + counters), driver-style bench lines
out=/root/repo/gpurun_out/r03z
mkdir -p $out
cd /root/repo
timeout 1700 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 900 bash scripts/collect_profiles.sh r03z_cfg2 > $out/collect_cfg2.log 2>&1; echo "collect cfg2 rc=$?"
for cfg in cfg1 cfg3 cfg4; do
  NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh r03z_$cfg --config $cfg > $out/collect_$cfg.log 2>&1; echo "collect $cfg rc=$?"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$?"
NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n2_gloo.json 2> $out/bench_n2_gloo.err; echo "bench n2 rc=$?"
timeout 600 python bench.py --gpus 1 --force-exchange --steps 200 --warmup 20 > $out/bench_rccl_world1.json 2> $out/bench_rccl_world1.err; echo "bench rccl1 rc=$?"
python scripts/train_hydrogen.py --out $out/train_cfg2_fp32.json > $out/train_cfg2_fp32.log 2>&1; tail -1 $out/train_cfg2_fp32.log | cut -c1-200
