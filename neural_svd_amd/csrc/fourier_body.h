// The fused path's feature stage as a device routine shared by its own kernel (fourier.hip) and by the backward chain
// kernel (pmlp_bwd.hip), which runs the NEXT batch's sampling + features as extra workgroups beside its latency-bound
// chain blocks (nsvd_operator_backward_evd_step_next).
#pragma once
#include "nsvd_kernels.h"

namespace nsvd_feat {

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

struct StencilArgs {
    const float* x;      // (B, D) coordinates, read when the sampler is off
    const float* fB;     // (D, m)
    float* phi;          // (B, 2m)
    float* phiTc;        // (2m, B) or null
    float* sctab;        // (D, 2, m) + (D, m)
    int B, m, D;
    float eps;
    NsvdSampler smp;
    float* xout;         // (B, D) drawn coordinates (sampler on)
};
constexpr int STAGE_FLOATS = 2 * FJ * (FB + 1);  // LDS of one tile: the transposed staging of sin and cos

// One tile = 64 frequencies x 32 base samples, by FT threads (tile (bx, by) of a (m / 64, B / 32) grid). The stand-alone
// kernel uses 1024 threads per block (16 waves, 4 per SIMD): each thread walks only 2 of the tile's 32 samples, so the
// ~1 us dependent chain of a double-accurate sincos is overlapped 4-fold instead of repeated 8 times; as guest blocks
// of the chain kernel (256 threads) a thread walks 8 samples in the shadow of the chain blocks.
template <int D, int FT>
__device__ __forceinline__ void stencil_tile(const StencilArgs& a, int bx, int by, float* lds) {
    const float* __restrict__ x = a.x;
    const float* __restrict__ fB = a.fB;
    float* __restrict__ phi = a.phi;
    float* __restrict__ phiTc = a.phiTc;
    float* __restrict__ sctab = a.sctab;
    const int B = a.B, m = a.m;
    const float eps = a.eps;
    const NsvdSampler& smp = a.smp;
    float* __restrict__ xout = a.xout;
    float (*ts)[FB + 1] = reinterpret_cast<float (*)[FB + 1]>(lds);                  // transposed staging of the centre rows
    float (*tc)[FB + 1] = reinterpret_cast<float (*)[FB + 1]>(lds + FJ * (FB + 1));
    const int tid = threadIdx.x;
    const int jl = tid & (FJ - 1);           // frequency within the tile (fastest: coalesced row stores)
    const int j = bx * FJ + jl;
    const int b0 = by * FB;
    const int F = 2 * m;
    const bool jok = j < m;
    float bj[D], sd[D], cd[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        bj[d] = jok ? fB[(size_t)d * m + j] : 0.f;
        sincos_d2f((double)eps * (double)bj[d], &sd[d], &cd[d]);
    }
    if (jok && by == 0 && tid < FJ) {  // per-frequency constants of the forward kernel, written once
        if (eps > 0.f) {  // stencil rows by angle addition: cos / sin of eps B_dj
#pragma unroll
            for (int d = 0; d < D; ++d) {
                sctab[(size_t)(2 * d) * m + j] = cd[d];
                sctab[(size_t)(2 * d + 1) * m + j] = sd[d];
                // cos(eps B_dj) - 1 = -2 sin^2(eps B_dj / 2), without the cancellation of cd - 1 (the bf16x3 forward
                // builds the stencil rows as centre + perturbation: pmlp_layer0_bf3.h)
                float sh, ch;
                sincos_d2f(0.5 * (double)eps * (double)bj[d], &sh, &ch);
                sctab[(size_t)(2 * D + d) * m + j] = -2.f * sh * sh;
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
    // the batch is DRAWN here (counter-based: every frequency block regenerates the same values, the first one stores
    // them for the epilogue and the backward). A wave walks RPW rows, all its lanes on the same row: lane k draws row k
    // ONCE (Philox + Box-Muller: ~200 instructions, six times the feature's own arithmetic) and the row loop reads it
    // with a scalar lane read, instead of every lane redoing it for every row.
    static_assert(FJ == 64 && FT % FJ == 0 && FB % (FT / FJ) == 0, "one wave per 64 frequencies, whole rows per wave");
    constexpr int NW = FT / FJ, RPW = FB / NW;
    const int wv = tid / FJ;
    float xmine[4] = {0.f, 0.f, 0.f, 0.f};
    if (smp.on && jl < RPW && b0 + wv + jl * NW < B) {
        const int b = b0 + wv + jl * NW;
        nsvd_sample_row(smp, b, D, xmine);
        if (bx == 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) xout[(size_t)b * D + d] = xmine[d];
        }
    }
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
        const int bl = wv + k * NW;
        const int b = b0 + bl;
        if (b >= B) break;
        float xr[4];
        if (smp.on) {
#pragma unroll
            for (int d = 0; d < D; ++d) xr[d] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xmine[d]), k));
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
        const int jg = bx * FJ + jj;
        if (jg < m && b0 + bl < B) {
            phiTc[(size_t)jg * B + b0 + bl] = ts[jj][bl];
            phiTc[(size_t)(m + jg) * B + b0 + bl] = tc[jj][bl];
        }
    }
}


}  // namespace nsvd_feat
