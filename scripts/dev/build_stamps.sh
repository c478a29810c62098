#!/bin/bash
# dev: diagnostic build with per-phase s_memtime stamps in the fused forward -> scripts/_diag/libnsvd_hip_stamps.so
#   [EXTRA="-DNSVD_HVAR_NOSP" SUF=_nosp] bash scripts/dev/build_stamps.sh: variant builds
#   NSVD_LIB_PATH=scripts/_diag/libnsvd_hip_stamps.so [NSVD_DEV_PATH=3] python scripts/dev/stamps.py
set -e
cd "$(dirname "$0")/../../neural_svd_amd/csrc"
mkdir -p ../../scripts/_diag
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fno-fast-math -ffp-contract=on -I../../include -DNSVD_FWD_STAMPS $EXTRA -c pmlp_fwd.hip -o ../../scripts/_diag/diag_fwd_stamps$SUF.o
objs=$(ls build/*.o | grep -v -e build/pmlp_fwd.o -e build/diag_)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scripts/_diag/libnsvd_hip_stamps$SUF.so $objs ../../scripts/_diag/diag_fwd_stamps$SUF.o
