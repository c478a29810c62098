#!/bin/bash
# the generic path's contractions (gemm_generic3 vectorised / gemm_generic2 pipelined / the round-4 kernel): parity tests
# on the generic path, then A/B timing
cd /root/repo
mkdir -p gpurun_out/r06gen
python -m pytest tests/test_hip_parity.py tests/test_dropin_gpu.py tests/test_kernel_apply_gpu.py -m gpu -x -q -k "generic or small or headline_shapes or backward or model_forward or exact or ragged or wide" 2>&1 | tail -8
echo "== vectorised kernel"; python scripts/dev/generic_time.py 2>&1 | grep hidden


[ -n "$AB" ] && { echo "== pipelined scalar kernel"; NSVD_GEMM_GENERIC3=0 python scripts/dev/generic_time.py 2>&1 | grep hidden; }
TOPN=14 scripts/gpu/prof_by_grid.sh r06gen256 /root/repo/scripts/dev/generic_time.py 256 | head -14
TOPN=14 scripts/gpu/prof_by_grid.sh r06gen64 /root/repo/scripts/dev/generic_time.py 64 | head -14
