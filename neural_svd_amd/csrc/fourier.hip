// Fourier feature map at the 1+2D stencil points, written feature-major (phiT[k][r]).
// Replaces reference examples/utils.py:139-140 evaluated at diff_ops.py:36-45's points.
#include <string.h>
#include "nsvd_kernels.h"

namespace {

__device__ __forceinline__ void sincos_d2f(double p, float* s, float* c) {
    // p reduced in double (|p| < 1e9: two-constant Cody-Waite is exact to ~1e-17 * n), polynomials in float
    const double n = rint(p * 0.63661977236758134308);
    double r = fma(n, -1.57079632679489655800e+00, p);
    r = fma(n, -6.12323399573676603587e-17, r);
    const float rf = (float)r;
    const float s2 = rf * rf;
    float ps = fmaf(s2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, s2, -1.6666654611e-1f);
    const float sn = fmaf(ps * s2, rf, rf);
    float pc = fmaf(s2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, s2, 4.166664568298827e-2f);
    const float cs = fmaf(pc * s2, s2, fmaf(-0.5f, s2, 1.0f));
    const int q = (int)((long long)n & 3);
    const float so = (q & 1) ? cs : sn;
    const float co = (q & 1) ? sn : cs;
    *s = (q & 2) ? -so : so;
    *c = ((q + 1) & 2) ? -co : co;
}

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

// Fused-path feature kernel: phi[r][k] for every stencil row (sample-major, k contiguous, ld = 2m) and,
// optionally, the feature-major copy of the centre rows phiT_c[k][b] for the weight-gradient GEMM.
// One workgroup = 64 frequencies x 32 base samples. Per (b, j) ONE sincos of the centre projection
// p = x_b . B_j evaluated in float64-accurate form (double projection + double Cody-Waite reduction), then
// the 2D shifted points by angle addition with d_i = eps * B_ij (sin d, cos d computed once per (i, j)):
//     sin(p +- d) = sin p cos d +- cos p sin d,   cos(p +- d) = cos p cos d -+ sin p sin d.
// This is the same function the reference evaluates (sin/cos((x +- eps e_i) . B)) but without the float32
// rounding of (x + eps) and of the projection, which are common-mode across the stencil here and would
// otherwise be amplified by 1/eps^2 in the finite-difference Laplacian; and it needs 1 + D instead of
// 1 + 2D sincos per (b, j).
constexpr int FJ = 64, FB = 32;

// 1024 threads per block (16 waves, 4 per SIMD): each thread walks only 2 of the tile's 32 samples, so the
// ~1 us dependent chain of a double-accurate sincos is overlapped 4-fold instead of repeated 8 times.
constexpr int FT = 1024;
template <int D>
__global__ void __launch_bounds__(FT) fourier_stencil_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ fB,
                                                              float* __restrict__ phi, float* __restrict__ phiTc,
                                                              float* __restrict__ sctab, int B, int m, float eps,
                                                              NsvdSampler smp, float* __restrict__ xout) {
    __shared__ float ts[FJ][FB + 1];  // transposed staging of the centre rows
    __shared__ float tc[FJ][FB + 1];
    const int tid = threadIdx.x;
    const int jl = tid & (FJ - 1);           // frequency within the tile (fastest: coalesced row stores)
    const int j = blockIdx.x * FJ + jl;
    const int b0 = blockIdx.y * FB;
    const int F = 2 * m;
    const bool jok = j < m;
    float bj[D], sd[D], cd[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        bj[d] = jok ? fB[(size_t)d * m + j] : 0.f;
        sincos_d2f((double)eps * (double)bj[d], &sd[d], &cd[d]);
    }
    if (jok && blockIdx.y == 0 && tid < FJ) {  // per-frequency constants of the forward kernel, written once
        if (eps > 0.f) {  // stencil rows by angle addition: cos / sin of eps B_dj
#pragma unroll
            for (int d = 0; d < D; ++d) {
                sctab[(size_t)(2 * d) * m + j] = cd[d];
                sctab[(size_t)(2 * d + 1) * m + j] = sd[d];
            }
        } else {          // exact-Laplacian jets: B_dj in the even slots, |B_j|^2 in slot 1
            float q = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                q = fmaf(bj[d], bj[d], q);
                sctab[(size_t)(2 * d) * m + j] = bj[d];
                if (d > 0) sctab[(size_t)(2 * d + 1) * m + j] = 0.f;
            }
            sctab[(size_t)m + j] = q;
        }
    }
    for (int bl = tid / FJ; bl < FB; bl += FT / FJ) {
        const int b = b0 + bl;
        if (b >= B) break;
        float xr[4];
        if (smp.on) {
            // the batch is DRAWN here (every frequency block regenerates the same counter-based values; the first
            // one stores them for the epilogue and the backward)
            nsvd_sample_row(smp, b, D, xr);
            if (blockIdx.x == 0 && jl == 0) {
#pragma unroll
                for (int d = 0; d < D; ++d) xout[(size_t)b * D + d] = xr[d];
            }
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) xr[d] = x[(size_t)b * D + d];
        }
        double p = 0.0;
#pragma unroll
        for (int d = 0; d < D; ++d) p = fma((double)xr[d], (double)bj[d], p);
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
    // phiT_c[k][b0 + bl]: 32 consecutive samples per frequency = one 128-B store
    const int bl = tid & (FB - 1);
    for (int jj = tid / FB; jj < FJ; jj += FT / FB) {
        const int jg = blockIdx.x * FJ + jj;
        if (jg < m && b0 + bl < B) {
            phiTc[(size_t)jg * B + b0 + bl] = ts[jj][bl];
            phiTc[(size_t)(m + jg) * B + b0 + bl] = tc[jj][bl];
        }
    }
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
    NsvdSampler smp;
    memset(&smp, 0, sizeof(smp));
    if (sampler) smp = *sampler;
    dim3 grid(nsvd_cdiv(m, FJ), nsvd_cdiv(B, FB));
    switch (D) {
        case 1: hipLaunchKernelGGL(fourier_stencil_kernel<1>, grid, dim3(FT), 0, s, x, fourier_B, phi, phiTc, sctab, B, m, eps, smp, xout); break;
        case 2: hipLaunchKernelGGL(fourier_stencil_kernel<2>, grid, dim3(FT), 0, s, x, fourier_B, phi, phiTc, sctab, B, m, eps, smp, xout); break;
        case 3: hipLaunchKernelGGL(fourier_stencil_kernel<3>, grid, dim3(FT), 0, s, x, fourier_B, phi, phiTc, sctab, B, m, eps, smp, xout); break;
        default: return NSVD_EUNSUPPORTED;
    }
    NSVD_CHECK_LAUNCH();
    return 0;
}
