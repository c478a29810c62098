#!/bin/bash
# dev: diagnostic build with per-block stamps in the wgrad kernel -> neural_svd_amd/libnsvd_hip_wgst.so
#   NSVD_LIB_PATH=neural_svd_amd/libnsvd_hip_wgst.so python scripts/dev_wgrad_stamps.py
set -e
cd "$(dirname "$0")/../neural_svd_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-fast-math -ffp-contract=on -I../../include -DNSVD_WG_STAMPS -c pmlp_bwd.hip -o build/diag_wgst.o
objs=$(ls build/*.o | grep -v -e build/pmlp_bwd.o -e build/diag_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnsvd_hip_wgst.so $objs build/diag_wgst.o
