#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py -m gpu -x -q 2>&1 | tail -3
for cfg in cfg2 cfg3; do BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03i_$cfg 2>&1 | tail -8; done
