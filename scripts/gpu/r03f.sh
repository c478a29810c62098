#!/bin/bash
# round 3, GPU call F: full GPU suite on the current tree + the default bench line + kernel stats / counters for cfg2, cfg3
out=/root/repo/gpurun_out/r03f
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -4 $out/pytest.log
timeout 900 bash scripts/collect_profiles.sh r03f_cfg2 > $out/collect_cfg2.log 2>&1; echo "collect cfg2 rc=$?"
timeout 900 bash scripts/collect_profiles.sh r03f_cfg3 --config cfg3 > $out/collect_cfg3.log 2>&1; echo "collect cfg3 rc=$?"
timeout 300 python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err; echo "bench rc=$?"
