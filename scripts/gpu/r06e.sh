#!/bin/bash
out=/root/repo/gpurun_out/r06e
mkdir -p $out
cd /root/repo
timeout 1500 python -m pytest tests/test_tower_gpu.py tests/test_cdk_step_gpu.py -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $out/pytest.log
