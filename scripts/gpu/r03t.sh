#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_multirank_gpu.py tests/test_kernel_apply_gpu.py -x -q -k "kernel_operator or fused_kernel or kernel_apply" 2>&1 | tail -5
python - <<'PY'
import torch, time
from neural_svd_amd.kernel_ops import FusedKernelTrainer, synthetic_psd_kernel
dev=torch.device('cuda:0')
op=synthetic_psd_kernel(10000,256,16,0,dev)
# per-rank shapes of the head-sharded step at world N, emulated: L/N heads on 8192 N rows
for N in (1,2,4,8):
    fk=FusedKernelTrainer(op,L=64//N,m=64,hidden=(128,128),batch_size=8192*N,lr=1e-4,seed=0)
    for _ in range(20): fk.step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(50): fk.step()
    torch.cuda.synchronize(); print(f"N={N}: {64//N} heads x {8192*N} rows: {(time.perf_counter()-t0)/50*1e3:.3f} ms/step")
    del fk
PY
