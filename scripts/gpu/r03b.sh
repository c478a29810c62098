#!/bin/bash
# round 3, GPU call B: the row-parallel moment phase of the chain kernel - parity subset, then per-kernel timings
out=/root/repo/gpurun_out/r03b
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py tests/test_multirank_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 $out/pytest.log
for cfg in cfg2 cfg3 cfg1; do
  BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03b_$cfg 2>&1 | tail -9
done
