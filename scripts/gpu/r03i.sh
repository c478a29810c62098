#!/bin/bash
cd /root/repo
NSVD_STAMPS_WS=1 NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_stamps.so python scripts/dev/stamps.py 2>&1 | grep -v amdgpu.ids | tail -10
