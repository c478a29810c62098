"""dev: cycles of block 0 / wave 0 of the bf16 contraction (gemm16.h).  python scripts/dev/gemm16_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import _lib, hip_ops as H
dev = "cuda:0"
M, N, K = 1024, 8192, 512
A = torch.randn(M, K, device=dev).bfloat16(); B = torch.randn(N, K, device=dev).bfloat16()
lib = _lib.load()
lib.nsvd_debug_g16_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 8)()
lib.nsvd_debug_g16_stamps(buf)  # arm
for _ in range(20): H.gemm_bf16(A, B, out_bf16=True)
torch.cuda.synchronize()
names = ["prologue", "step_first_half", "step_barrier", "step_second_half", "epilogue", "nk"]
lib.nsvd_debug_g16_stamps(buf)
print("T,T (1024, 8192, 512) bf16 out:", {n: int(buf[i]) for i, n in enumerate(names)})
A2 = torch.randn(1024, 512, device=dev).bfloat16(); B2 = torch.randn(1024, 8192, device=dev).bfloat16()
for _ in range(20): H.gemm_bf16(A2, B2, a_kstrided=True, b_kstrided=True)
torch.cuda.synchronize()
lib.nsvd_debug_g16_stamps(buf)
print("S,S (512, 8192, 1024) f32 out:", {n: int(buf[i]) for i, n in enumerate(names)})
A3 = torch.randn(1024, 512, device=dev).bfloat16(); B3 = torch.randn(512, 8192, device=dev).bfloat16()
for _ in range(20): H.gemm_bf16(A3, B3, b_kstrided=True, out_bf16=True)
torch.cuda.synchronize()
lib.nsvd_debug_g16_stamps(buf)
print("T,S (1024, 8192, 512) bf16 out:", {n: int(buf[i]) for i, n in enumerate(names)})
