#!/bin/bash
# round 3, GPU call D: stamps of the tile-pipelined hidden layers + parity of the default build + kernel timings
cd /root/repo
NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_stamps.so python scripts/dev/stamps.py 2>&1 | grep -v "amdgpu.ids" | tail -10
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
for cfg in cfg2 cfg3; do BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03d_$cfg 2>&1 | tail -8; done
