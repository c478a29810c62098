#!/bin/bash
# round 3, GPU call A: full GPU suite, driver-style bench lines (N = 1 and the self-launched N = 2 on one device over
# gloo), rocprofv3 evidence for cfg3, cfg1, cfg5, cfg4
out=/root/repo/gpurun_out/r03a
mkdir -p $out
cd /root/repo
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench n1 rc=$?"
NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_n2_gloo.json 2> $out/bench_n2_gloo.err; echo "bench n2 rc=$?"
for cfg in cfg3 cfg1 cfg5 cfg4; do
  timeout 900 bash scripts/collect_profiles.sh r03a_$cfg --config $cfg > $out/collect_$cfg.log 2>&1; echo "collect $cfg rc=$?"
done
