#!/bin/bash
# dev: diagnostic build with per-block stamps in the wgrad kernel -> scripts/_diag/libnsvd_hip_wgst.so
#   NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_wgst.so python scripts/dev/wgrad_stamps.py
set -e
cd "$(dirname "$0")/../../neural_svd_amd/csrc"
mkdir -p ../../scripts/_diag
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-fast-math -ffp-contract=on -I../../include -DNSVD_WG_STAMPS $EXTRA -c pmlp_bwd.hip -o ../../scripts/_diag/diag_wgst.o
objs=$(ls build/*.o | grep -v -e build/pmlp_bwd.o -e build/diag_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/_diag/libnsvd_hip_wgst$SUF.so $objs ../../scripts/_diag/diag_wgst.o
