#!/bin/bash
cd /root/repo
timeout 1500 python -m pytest tests/test_multirank_gpu.py tests/test_tower_gpu.py tests/test_cdk_step_gpu.py -x -q -k "hidden_width or tower or cdk_step" 2>&1 | tail -25
