#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_dropin_gpu.py -x -q -k "reference_style or train_operator" 2>&1 | tail -25
