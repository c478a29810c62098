#!/bin/bash
# round 3, GPU call J (final tree): full GPU suite, evidence for profiles/ (kernel stats + counters), driver-style bench
# lines, end-to-end training records (eigenvalue error after the reference's schedules)
out=/root/repo/gpurun_out/r03j
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
timeout 900 bash scripts/collect_profiles.sh r03j_cfg2 > $out/collect_cfg2.log 2>&1; echo "collect cfg2 rc=$?"
timeout 900 bash scripts/collect_profiles.sh r03j_cfg3 --config cfg3 > $out/collect_cfg3.log 2>&1; echo "collect cfg3 rc=$?"
for cfg in cfg1 cfg4 cfg5; do
  NSVD_PROFILE_PMC=0 timeout 600 bash scripts/collect_profiles.sh r03j_$cfg --config $cfg > $out/collect_$cfg.log 2>&1; echo "collect $cfg rc=$?"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$?"
NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n2_gloo.json 2> $out/bench_n2_gloo.err; echo "bench n2 rc=$?"
python scripts/train_hydrogen.py --out $out/train_cfg2_fp32.json > $out/train_cfg2_fp32.log 2>&1; tail -1 $out/train_cfg2_fp32.log | cut -c1-200
python scripts/train_hydrogen.py --sequential --out $out/train_cfg2_seq.json > $out/train_cfg2_seq.log 2>&1; tail -1 $out/train_cfg2_seq.log | cut -c1-200
python scripts/train_hydrogen.py --path bf16x3 --out $out/train_cfg2_bf16x3.json > $out/train_cfg2_bf16x3.log 2>&1; tail -1 $out/train_cfg2_bf16x3.log | cut -c1-200
python scripts/train_hydrogen.py --laplacian-eps 0 --out $out/train_cfg2_exact.json > $out/train_cfg2_exact.log 2>&1; tail -1 $out/train_cfg2_exact.log | cut -c1-200
python scripts/train_hydrogen.py --problem oscillator --evals 100000 --out $out/train_cfg3_oscillator.json > $out/train_cfg3_oscillator.log 2>&1; tail -1 $out/train_cfg3_oscillator.log | cut -c1-200
python scripts/train_hydrogen.py --problem oscillator --neigs 55 --batch-size 512 --evals 100000 --out $out/train_osc_L55_B512.json > $out/train_osc_L55_B512.log 2>&1; tail -1 $out/train_osc_L55_B512.log | cut -c1-200
python scripts/train_hydrogen.py --problem oscillator --neigs 55 --batch-size 4096 --evals 100000 --out $out/train_osc_L55_B4096.json > $out/train_osc_L55_B4096.log 2>&1; tail -1 $out/train_osc_L55_B4096.log | cut -c1-200
