// Fourier feature map at the 1+2D stencil points, written feature-major (phiT[k][r]).
// Replaces reference examples/utils.py:139-140 evaluated at diff_ops.py:36-45's points.
#include "nsvd_kernels.h"

namespace {

// one thread per (stencil row r, frequency j): sin and cos of the same projection (nsvd_sincos:
// ~1 ulp - projections reach tens of radians and the FD Laplacian amplifies feature error by 1/eps^2).
__global__ void __launch_bounds__(256) fourier_kernel(const float* __restrict__ x, const float* __restrict__ fB,
                                                      float* __restrict__ phiT, int B, int D, int m, float eps,
                                                      int nst, int ldr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int R = nst * B;
    if (r >= R) return;
    const int e = r / B;
    const int b = r - e * B;
    float proj = 0.f;
    for (int d = 0; d < D; ++d) {
        const float xc = nsvd_stencil_coord(x[(size_t)b * D + d], d, e, eps);
        proj = fmaf(xc, fB[(size_t)d * m + j], proj);
    }
    float s, c;
    nsvd_sincos(proj, &s, &c);
    phiT[(size_t)j * ldr + r] = s;
    phiT[(size_t)(m + j) * ldr + r] = c;
}

// sample-major variant for the fused MFMA forward: phi[r][k], k contiguous (ld = 2m), one thread per
// (frequency j, stencil row r), j fastest so both stores of a wave are 256-B contiguous.
__global__ void __launch_bounds__(256) fourier_rows_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ fB, float* __restrict__ phi,
                                                           int B, int D, int m, float eps, int nst) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (j >= m) return;
    const int e = r / B;
    const int b = r - e * B;
    float proj = 0.f;
    for (int d = 0; d < D; ++d) {
        const float xc = nsvd_stencil_coord(x[(size_t)b * D + d], d, e, eps);
        proj = fmaf(xc, fB[(size_t)d * m + j], proj);
    }
    float s, c;
    nsvd_sincos(proj, &s, &c);
    phi[(size_t)r * (2 * m) + j] = s;
    phi[(size_t)r * (2 * m) + m + j] = c;
}

}  // namespace

extern "C" int nsvd_fourier_features(const float* x, const float* fourier_B, float* phiT, int B, int D, int m,
                                     float eps, int nstencil, int ldr, void* stream) {
    if (!x || !fourier_B || !phiT || B <= 0 || D <= 0 || m <= 0) return NSVD_EINVAL;
    if (nstencil != 1 && nstencil != 1 + 2 * D) return NSVD_EINVAL;
    if (ldr < nstencil * B) return NSVD_EINVAL;
    const int R = nstencil * B;
    dim3 grid(nsvd_cdiv(R, 256), m);
    hipLaunchKernelGGL(fourier_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, fourier_B, phiT, B, D, m, eps,
                       nstencil, ldr);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_fourier_rows(const float* x, const float* fourier_B, float* phi, int B, int D, int m, float eps,
                      int nstencil, hipStream_t s) {
    const int R = nstencil * B;
    dim3 grid(nsvd_cdiv(m, 256), R);
    hipLaunchKernelGGL(fourier_rows_kernel, grid, dim3(256), 0, s, x, fourier_B, phi, B, D, m, eps, nstencil);
    NSVD_CHECK_LAUNCH();
    return 0;
}
