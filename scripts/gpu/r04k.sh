#!/bin/bash
# round 4: even / odd stencil form in the NATIVE float32 kernel - accuracy, parity tests, kernel time A/B against
# the previous library (OLD_LIB = a build of the parent commit, neural_svd_amd/_ab/)
out=/root/repo/gpurun_out/r04k
mkdir -p $out
cd /root/repo
timeout 300 python scripts/dev/bf3_check.py > $out/bf3_check.log 2>&1; echo "bf3_check rc=$?"; grep -E "RECORD|forward|vs fp32" $out/bf3_check.log | cut -c1-900
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_spectrum_parity_gpu.py tests/test_dropin_gpu.py -x -q > $out/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest_sel.log
OLD_LIB=/root/repo/neural_svd_amd/_ab/libnsvd_hip_old.so bash scripts/gpu/r04j.sh
