#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py tests/test_kernel_apply_gpu.py tests/test_multirank_gpu.py -m gpu -x -q 2>&1 | tail -4
for cfg in cfg3 cfg4; do BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03l_$cfg 2>&1 | tail -8; done
