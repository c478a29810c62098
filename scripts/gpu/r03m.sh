#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_bench_launch.py -m gpu -x -q 2>&1 | tail -5
