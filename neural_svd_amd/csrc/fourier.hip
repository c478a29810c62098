// Fourier feature map at the 1+2D stencil points, written feature-major (phiT[k][r]).
// Replaces reference examples/utils.py:139-140 evaluated at diff_ops.py:36-45's points.
#include <string.h>
#include "nsvd_kernels.h"
#include "fourier_body.h"

using nsvd_feat::sincos_d2f;
using nsvd_feat::FJ;
using nsvd_feat::FB;

namespace {

// generic path: one thread per (stencil row r, frequency j), feature-major output phiT[k][r]
__global__ void __launch_bounds__(256) fourier_kernel(const float* __restrict__ x, const float* __restrict__ fB,
                                                      float* __restrict__ phiT, int B, int D, int m, float eps,
                                                      int nst, int ldr) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    const int R = nst * B;
    if (r >= R) return;
    const int e = r / B;
    const int b = r - e * B;
    // shifted coordinate and projection in double: their float32 rounding differs from stencil point to
    // stencil point and would be amplified by 1/eps^2 in the finite-difference Laplacian
    double proj = 0.0;
    for (int d = 0; d < D; ++d) {
        double xc = (double)x[(size_t)b * D + d];
        if (e > 0 && ((e - 1) >> 1) == d) xc += ((e - 1) & 1) ? -(double)eps : (double)eps;
        proj = fma(xc, (double)fB[(size_t)d * m + j], proj);
    }
    float s, c;
    sincos_d2f(proj, &s, &c);
    phiT[(size_t)j * ldr + r] = s;
    phiT[(size_t)(m + j) * ldr + r] = c;
}

// the same with the shifted rows in EVEN / ODD form: row block 1 + 2 d holds phi's even perturbation
// along direction d, [s (cos t - 1), c (cos t - 1)] with t = eps B_dj, block 2 + 2 d the odd one, [c sin t, -s sin t]:
// phi(x +- eps e_d) = phi + even +- odd exactly (angle addition), products in double
__global__ void __launch_bounds__(256) fourier_evenodd_kernel(const float* __restrict__ x, const float* __restrict__ fB,
                                                              float* __restrict__ phiT, int B, int D, int m, float eps,
                                                              int ldr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (b >= B) return;
    double proj = 0.0;
    for (int d = 0; d < D; ++d) proj = fma((double)x[(size_t)b * D + d], (double)fB[(size_t)d * m + j], proj);
    float s, c;
    sincos_d2f(proj, &s, &c);
    phiT[(size_t)j * ldr + b] = s;
    phiT[(size_t)(m + j) * ldr + b] = c;
    for (int d = 0; d < D; ++d) {
        const double t = (double)eps * (double)fB[(size_t)d * m + j];
        const double sh = sin(0.5 * t), cm = -2.0 * sh * sh, sd = sin(t);
        const size_t re = (size_t)(1 + 2 * d) * B + b, ro = (size_t)(2 + 2 * d) * B + b;
        phiT[(size_t)j * ldr + re] = (float)((double)s * cm);
        phiT[(size_t)(m + j) * ldr + re] = (float)((double)c * cm);
        phiT[(size_t)j * ldr + ro] = (float)((double)c * sd);
        phiT[(size_t)(m + j) * ldr + ro] = (float)(-(double)s * sd);
    }
}

constexpr int FT = 1024;
template <int D>
__global__ void __launch_bounds__(FT) fourier_stencil_kernel(nsvd_feat::StencilArgs a) {
    __shared__ float lds[nsvd_feat::STAGE_FLOATS];
    nsvd_feat::stencil_tile<D, FT>(a, blockIdx.x, blockIdx.y, lds);
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

int nsvd_fourier_features_evenodd(const float* x, const float* fourier_B, float* phiT, int B, int D, int m, float eps,
                                  int ldr, hipStream_t s) {
    if (!x || !fourier_B || !phiT || B <= 0 || D <= 0 || m <= 0 || ldr < (1 + 2 * D) * B) return NSVD_EINVAL;
    hipLaunchKernelGGL(fourier_evenodd_kernel, dim3(nsvd_cdiv(B, 256), m), dim3(256), 0, s, x, fourier_B, phiT, B, D, m,
                       eps, ldr);
    NSVD_CHECK_LAUNCH();
    return 0;
}

namespace {
// centre features for any input dimension (runtime D): same tile shape and double-accurate projection as above
__global__ void __launch_bounds__(FT) fourier_plain_kernel(const float* __restrict__ x, const float* __restrict__ fB,
                                                           float* __restrict__ phi, float* __restrict__ phiTc, int B,
                                                           int D, int m) {
    __shared__ float ts[FJ][FB + 1];
    __shared__ float tc[FJ][FB + 1];
    const int tid = threadIdx.x;
    const int jl = tid & (FJ - 1);
    const int j = blockIdx.x * FJ + jl;
    const int b0 = blockIdx.y * FB;
    const int F = 2 * m;
    const bool jok = j < m;
    for (int bl = tid / FJ; bl < FB; bl += FT / FJ) {
        const int b = b0 + bl;
        if (b >= B) break;
        double p = 0.0;
        if (jok)
            for (int d = 0; d < D; ++d) p = fma((double)x[(size_t)b * D + d], (double)fB[(size_t)d * m + j], p);
        float s0, c0;
        sincos_d2f(p, &s0, &c0);
        if (jok) {
            float* row = phi + (size_t)b * F;
            row[j] = s0;
            row[m + j] = c0;
        }
        ts[jl][bl] = s0;
        tc[jl][bl] = c0;
    }
    if (!phiTc) return;
    __syncthreads();
    const int bl = tid & (FB - 1);
    for (int jj = tid / FB; jj < FJ; jj += FT / FB) {
        const int jg = blockIdx.x * FJ + jj;
        if (jg < m && b0 + bl < B) {
            phiTc[(size_t)jg * B + b0 + bl] = ts[jj][bl];
            phiTc[(size_t)(m + jg) * B + b0 + bl] = tc[jj][bl];
        }
    }
}

__global__ void __launch_bounds__(256) sample_kernel(NsvdSampler smp, float* __restrict__ x, int B, int D) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    float xr[4];
    nsvd_sample_row(smp, b, D, xr);
    for (int d = 0; d < D; ++d) x[(size_t)b * D + d] = xr[d];
}
}  // namespace

int nsvd_fourier_plain(const float* x, const float* fourier_B, float* phi, float* phiTc, int B, int D, int m,
                       hipStream_t s) {
    hipLaunchKernelGGL(fourier_plain_kernel, dim3(nsvd_cdiv(m, FJ), nsvd_cdiv(B, FB)), dim3(FT), 0, s, x, fourier_B, phi,
                       phiTc, B, D, m);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_sample_launch(const NsvdSampler& smp, float* x, int B, int D, hipStream_t s) {
    if (D < 1 || D > 4) return NSVD_EUNSUPPORTED;
    hipLaunchKernelGGL(sample_kernel, dim3(nsvd_cdiv(B, 256)), dim3(256), 0, s, smp, x, B, D);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_fourier_stencil(const float* x, const float* fourier_B, float* phi, float* phiTc, float* sctab, int B, int D,
                         int m, float eps, const NsvdSampler* sampler, float* xout, hipStream_t s) {
    nsvd_feat::StencilArgs a;
    memset(&a, 0, sizeof(a));
    if (sampler) a.smp = *sampler;
    a.x = x; a.fB = fourier_B; a.phi = phi; a.phiTc = phiTc; a.sctab = sctab; a.B = B; a.m = m; a.D = D; a.eps = eps;
    a.xout = xout;
    dim3 grid(nsvd_cdiv(m, FJ), nsvd_cdiv(B, FB));
    switch (D) {
        case 1: hipLaunchKernelGGL(fourier_stencil_kernel<1>, grid, dim3(FT), 0, s, a); break;
        case 2: hipLaunchKernelGGL(fourier_stencil_kernel<2>, grid, dim3(FT), 0, s, a); break;
        case 3: hipLaunchKernelGGL(fourier_stencil_kernel<3>, grid, dim3(FT), 0, s, a); break;
        default: return NSVD_EUNSUPPORTED;
    }
    NSVD_CHECK_LAUNCH();
    return 0;
}
