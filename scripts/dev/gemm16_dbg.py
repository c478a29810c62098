"""dev: the fwd1-shaped contraction alone, many launches (for rocprofv3 --stats under NSVD_G16_DBG = 0 / 1 / 2)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
dev = "cuda:0"
M, N, K = 1024, 8192, 512
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
for _ in range(300): H.gemm_bf16(A, B, out_bf16=True)
A2 = torch.randn(1024, 512, device=dev).bfloat16(); B2 = torch.randn(1024, 8192, device=dev).bfloat16()
for _ in range(300): H.gemm_bf16(A2, B2, a_kstrided=True, b_kstrided=True)
torch.cuda.synchronize()
