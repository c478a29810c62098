"""dev: nsvd_gemm_bf16 at the five shapes of a mixed-precision tower (B = 1024, 512 -> 8192 -> 512), us per launch
and TFLOP/s, next to torch.matmul (hipBLASLt) on the same bf16 operands."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from neural_svd_amd import hip_ops as H
dev = "cuda:0"
shapes = [("fwd1 Y1=X W1^T", 1024, 8192, 512, False, False, True, 1), ("fwd1 f32 out", 1024, 8192, 512, False, False, False, 1),
          ("fwd2 split-K 8", 1024, 512, 8192, False, False, False, 8),
          ("dW2 SS", 512, 8192, 1024, True, True, False, 1), ("dA1 TS", 1024, 8192, 512, False, True, True, 1),
          ("dW1 SS", 8192, 512, 1024, True, True, False, 1)]
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for name, M, N, K, a_s, b_s, o16, S in shapes:
    A = torch.randn((K, M) if a_s else (M, K), device=dev).bfloat16()
    B = torch.randn((K, N) if b_s else (N, K), device=dev).bfloat16()
    t = timeit(lambda: H.gemm_bf16(A, B, a_kstrided=a_s, b_kstrided=b_s, out_bf16=o16, slices=S))
    At = A.t() if a_s else A
    Bt = B if b_s else B.t()
    tt = timeit(lambda: torch.matmul(At, Bt))
    fl = 2.0 * M * N * K
    print(f"{name:18s} M={M} N={N} K={K}: {t * 1e6:7.1f} us {fl / t / 1e12:7.1f} TF/s | torch.matmul bf16 {tt * 1e6:7.1f} us {fl / tt / 1e12:7.1f} TF/s")
