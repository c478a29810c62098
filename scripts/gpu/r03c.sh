#!/bin/bash
# round 3, GPU call C: tile-pipelined hidden layers of the forward kernel - parity, timings, stamps
out=/root/repo/gpurun_out/r03c
mkdir -p $out
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py tests/test_spectrum_parity_gpu.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $out/pytest.log
for cfg in cfg2 cfg3; do
  BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03c_$cfg 2>&1 | tail -9
done
NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_stamps.so python scripts/dev/stamps.py 2>&1 | tail -14
