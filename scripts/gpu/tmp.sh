#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_tower_gpu.py tests/test_cdk_step_gpu.py tests/test_cdk_gpu.py -x -q 2>&1 | tail -15
BENCH_ARGS="--config cfg5 --amp" bash scripts/dev/ab.sh tmp_cfg5amp 2>&1 | grep -v "^[WE]2026" | tail -12
