#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
for cfg in cfg1 cfg2; do BENCH_ARGS="--config $cfg" bash scripts/dev/ab.sh r03n_$cfg 2>&1 | tail -8; done
