#!/bin/bash
# dev: diagnostic build with per-block stamps in the wgrad kernel -> neural_svd_amd/libnsvd_hip_wgst.so
set -e
cd "$(dirname "$0")/../neural_svd_amd/csrc"
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-fast-math -ffp-contract=on -I../../include -DNSVD_WG_STAMPS -c pmlp_fused.hip -o build/pmlp_wgst.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnsvd_hip_wgst.so build/fourier.o build/gemm_generic.o build/fd_epilogue.o build/evd_loss.o build/optimizer.o build/spectrum.o build/pmlp_wgst.o build/cdk_loss.o build/kernel_apply.o build/operator_api.o
