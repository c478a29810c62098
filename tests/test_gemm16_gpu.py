"""GPU parity of nsvd_gemm_bf16 (csrc/gemm16.h, gemm16b.h) - the bf16-MFMA contraction of the mixed-precision CDK towers - against
float64 products of the SAME bfloat16 operand values (oracle arithmetic: numpy / torch float64 on the CPU), in every
operand form the towers use, at the towers' five shapes (BASELINE configs[4]: B = 1024, 512 -> 8192 -> 512)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand_bf16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(shape, generator=g)).to(torch.bfloat16)


def _check(M, N, K, a_s, b_s, out16, slices=1, bias=False, seed=0, rows=64):
    from neural_svd_amd import hip_ops as H
    A = _rand_bf16((K, M) if a_s else (M, K), seed)
    B = _rand_bf16((K, N) if b_s else (N, K), seed + 1)
    bv = torch.randn(N, generator=torch.Generator().manual_seed(seed + 2)) if bias else None
    out = H.gemm_bf16(A.to(DEV), B.to(DEV), bias=bv.to(DEV) if bias else None, a_kstrided=a_s, b_kstrided=b_s,
                      out_bf16=out16, slices=slices, want_sumsq=True)
    C, ss = out
    torch.cuda.synchronize()
    C = C.float().cpu().double()
    # float64 oracle on sampled rows (every column): exact products of the bf16 values
    A64 = (A.double().t() if a_s else A.double())  # (M, K)
    B64 = (B.double().t() if b_s else B.double())  # (N, K)
    idx = torch.randperm(M, generator=torch.Generator().manual_seed(seed + 3))[:rows]
    Ks = K // slices
    for s in range(slices):
        want = A64[idx, s * Ks:(s + 1) * Ks] @ B64[:, s * Ks:(s + 1) * Ks].t()
        if bias:
            want = want + bv.double()
        got = C[idx] if slices == 1 else C[s][idx]
        scale = float(want.abs().max())
        tol = (2.0 ** -8 if out16 else 2e-6 * np.sqrt(Ks)) * scale  # bf16 output rounding / fp32 accumulation
        assert float((got - want).abs().max()) <= tol, (s, float((got - want).abs().max()), tol)
    # per-tile sums of squares of what was stored
    Cs = C if slices > 1 else C[None]
    tiles = (Cs.reshape(slices, M // 256, 256, N // 128, 128) ** 2).sum(dim=(2, 4)).reshape(-1)
    assert torch.allclose(ss.cpu().double(), tiles, rtol=(2e-2 if out16 else 1e-4))


@pytest.mark.parametrize("M,N,K,a_s,b_s,out16,slices,bias", [
    (256, 128, 64, False, False, False, 1, False),     # one tile, one K step
    (256, 128, 128, False, False, False, 1, True),     # two K steps (the ring's prologue only)
    (256, 128, 320, True, True, False, 1, False),      # five K steps: the ring wraps
    (512, 256, 256, False, True, True, 1, True),
    (1024, 8192, 512, False, False, True, 1, True),    # Y1 = X W1^T + b1
    (1024, 512, 8192, False, False, False, 8, False),  # Y2 = A1 W2^T, split-K
    (512, 8192, 1024, True, True, False, 1, False),    # dW2 = dY2^T A1
    (1024, 8192, 512, False, True, True, 1, False),    # dA1 = dY2 W2
    (8192, 512, 1024, True, True, False, 1, False),    # dW1 = dY1^T X
    # 512 tiles and more: the two-workgroups-per-CU form of the kernel (gemm16b.h: K steps of 32, three-stage ring)
    (2048, 8192, 64, False, False, True, 1, True),     # two K steps of 32
    (2048, 8192, 320, False, False, True, 1, True),    # ten: the ring wraps
    (2048, 8192, 320, False, True, True, 1, False),
    (2048, 8192, 320, True, True, False, 1, False),    # float32 tile out in two passes
    (2048, 8192, 192, False, False, False, 1, False),
    (2048, 4096, 384, True, True, True, 2, False),     # split-K slices in this form
])
def test_gemm_bf16_forms(M, N, K, a_s, b_s, out16, slices, bias):
    _check(M, N, K, a_s, b_s, out16, slices, bias)


def test_gemm_bf16_identity_operand_catches_layout_swaps():
    """A = I (as T and as S operand) against an ASYMMETRIC B: the output must be B^T's rows exactly (bf16 values are
    exact in float32) - a swapped row / column map or fragment k-order cannot pass."""
    from neural_svd_amd import hip_ops as H
    M = K = 256
    N = 128
    eye = torch.eye(M, dtype=torch.bfloat16, device=DEV)
    Bm = _rand_bf16((N, K), 5).to(DEV)
    for a_s in (False, True):
        for b_s in (False, True):
            if a_s and not b_s:
                continue
            C = H.gemm_bf16(eye, Bm.t().contiguous() if b_s else Bm, a_kstrided=a_s, b_kstrided=b_s)
            assert torch.equal(C, Bm.float().t()), (a_s, b_s)


def test_gemm_bf16_refuses_bad_shapes():
    from neural_svd_amd import hip_ops as H
    from neural_svd_amd._lib import NsvdError
    A = torch.zeros(256, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(NsvdError):
        H.gemm_bf16(A, torch.zeros(100, 64, dtype=torch.bfloat16, device=DEV))  # N not a multiple of 128
    with pytest.raises(NsvdError):
        H.gemm_bf16(torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV),
                    torch.zeros(128, 64, dtype=torch.bfloat16, device=DEV), a_kstrided=True)  # (S, T) form not built


def test_to_bf16_rounds_to_nearest_even():
    from neural_svd_amd import hip_ops as H
    x = torch.randn(4096, generator=torch.Generator().manual_seed(0))
    x[:8] = torch.tensor([1.0, 1.00390625, 1.01171875, -1.00390625, 3.3895314e38, 1e-40, 0.0, -0.0])
    assert torch.equal(H.to_bf16(x.to(DEV)).cpu(), x.to(torch.bfloat16))
