#!/bin/bash
# the pipelined generic contraction (gemm_generic2): parity tests on the generic path, then A/B timing against the old kernel
cd /root/repo
mkdir -p gpurun_out/r06gen
python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py tests/test_kernel_apply_gpu.py -m gpu -x -q -k "generic or small or headline_shapes or backward or model_forward or exact or ragged or wide" 2>&1 | tail -8
echo "== new kernel"; python scripts/dev/generic_time.py 2>&1 | grep hidden
echo "== old kernel"; NSVD_GEMM_GENERIC2=0 python scripts/dev/generic_time.py 2>&1 | grep hidden
