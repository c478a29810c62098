#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_kernel_apply_gpu.py -x -q -k "kernel_operator or fused_kernel" 2>&1 | tail -15
timeout 600 env NSVD_FORCE_DEVICE=0 NSVD_DIST_BACKEND=gloo python bench.py --config cfg4 --gpus 2 --steps 20 --warmup 5 2>&1 | tail -3 | cut -c 1-2500
timeout 600 python bench.py --config cfg4 --steps 50 --warmup 5 2>&1 | tail -1 | cut -c 1-600
