#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_multirank_gpu.py -x -q 2>&1 | tail -8
