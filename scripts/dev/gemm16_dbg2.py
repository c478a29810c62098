"""dev: a 512-tile contraction (two tiles per persistent workgroup) for rocprofv3 --stats under NSVD_G16_DBG"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
dev = "cuda:0"
M, N, K = 2048, 8192, 512
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
for _ in range(300): H.gemm_bf16(A, B, out_bf16=True)
torch.cuda.synchronize()
