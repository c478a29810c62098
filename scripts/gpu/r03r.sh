#!/bin/bash
cd /root/repo
timeout 1800 python -m pytest tests/test_bench_launch.py -x -q 2>&1 | tail -15
