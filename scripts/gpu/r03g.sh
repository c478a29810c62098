#!/bin/bash
# round 3, GPU call G: guest feature workgroups in the chain kernel - parity + timings
cd /root/repo
timeout 900 python -m pytest tests/test_dropin_gpu.py tests/test_hip_parity.py tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -5
for cfg in cfg2 cfg3 cfg1; do BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03g_$cfg 2>&1 | tail -8; done
