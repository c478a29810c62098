// Generic strided, batched fp32 GEMM for shapes the fused MFMA kernels do not cover
// (hidden widths that are not 128, ragged batches). LDS-tiled 64 x 64 x 16; the products on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: a wave owns one 32 x 32 quadrant of the tile, one MFMA and two 4-byte LDS reads per pair of
// k) since round 4 - rounds 1-3 multiplied on the vector ALU (4 x 4 register tiles, 36 TFLOP/s).
// Used for: ParallelMLP layers (reference mlp.py:204-221), their data gradients and weight
// gradients (what autograd derives for those einsums).
#include <stdlib.h>
#include "nsvd_kernels.h"

namespace {

constexpr int TM = 64, TN = 64, TK = 16, PAD = 4;

__global__ void __launch_bounds__(256) gemm_generic_kernel(NsvdGemm g) {
    __shared__ float As[TK][TM + PAD];
    __shared__ float Bs[TK][TN + PAD];
    const int bz = blockIdx.z;
    const float* A = g.A + (size_t)bz * g.bA;
    const float* Bm = g.B + (size_t)bz * g.bB;
    float* C = g.C + (size_t)bz * g.bC;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;  // this wave's quadrant: rows 32 wm .., columns 32 wn ..
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    f32x16_t acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool a_kfast = (g.sAk == 1);
    const bool b_nfast = (g.sBn == 1);
    for (int k0 = 0; k0 < g.K; k0 += TK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int mm, kk;
            if (a_kfast) { kk = t & 15; mm = (t >> 4) + 16 * i; }
            else         { mm = t & 63; kk = (t >> 6) + 4 * i; }
            const int gm = m0 + mm, gk = k0 + kk;
            float v = 0.f;
            if (gm < g.M && gk < g.K) v = A[(size_t)gm * g.sAm + (size_t)gk * g.sAk];
            As[kk][mm] = v;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int nn, kk;
            if (b_nfast) { nn = t & 63; kk = (t >> 6) + 4 * i; }
            else         { kk = t & 15; nn = (t >> 4) + 16 * i; }
            const int gn = n0 + nn, gk = k0 + kk;
            float v = 0.f;
            if (gn < g.N && gk < g.K) {
                v = Bm[(size_t)gk * g.sBk + (size_t)gn * g.sBn];
                if (g.softplus_b) {
                    if (g.eo_cols > 0 && gn >= g.eo_cols) {
                        const int e = gn / g.eo_cols, bcol = gn - e * g.eo_cols, dd = (e - 1) >> 1;
                        const float* zr = Bm + (size_t)gk * g.sBk;
                        const float z0 = zr[(size_t)bcol * g.sBn];
                        const float zE = zr[(size_t)((1 + 2 * dd) * g.eo_cols + bcol) * g.sBn];
                        const float zO = zr[(size_t)((2 + 2 * dd) * g.eo_cols + bcol) * g.sBn];
                        float ev, od;
                        nsvd_softplus_evenodd(z0, zE, zO, &ev, &od);
                        v = ((e - 1) & 1) ? od : ev;
                    } else {
                        v = nsvd_softplus(v);
                    }
                }
            }
            Bs[kk][nn] = v;
        }
        __syncthreads();
        // lane (li, hi) feeds row / column li of the quadrant at k = kk + hi: D[i][j] += sum_k A[i][k] B[k][j]
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + hi][32 * wm + li], Bs[kk + hi][32 * wn + li], acc, 0, 0, 0);
        __syncthreads();
    }
    const float* bias = g.bias ? g.bias + (size_t)bz * g.bBias : nullptr;
    const float* Z = g.Z ? g.Z + (size_t)bz * g.bZ : nullptr;
    // accumulator register r of lane (li, hi): row 8 (r / 4) + 4 hi + r % 4, column li of the quadrant
    const int gn = n0 + 32 * wn + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gm = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (gm >= g.M || gn >= g.N) continue;
        const float bi = bias ? bias[gm] : 0.f;
        float v = acc[r] + ((g.eo_cols > 0 && gn >= g.eo_cols) ? 0.f : bi);
        if (g.sigmoid_mul) v *= nsvd_sigmoid(Z[(size_t)gm * g.sZm + gn]);
        C[(size_t)gm * g.sCm + gn] = v;
    }
}

// The same contraction on 64 x 128 tiles with the NEXT K tile's global loads in flight under the current tile's MFMAs
// (round 6): a wave owns 32 rows x 64 columns (two 32 x 32 accumulators fed by ONE A value: three 4-byte LDS reads per
// two MFMAs instead of four), the LDS tiles are double buffered (one workgroup barrier per 16 k instead of two per 16),
// the staging values of tile t + 1 are requested before tile t is multiplied and written behind it. Takes every launch
// except the stencil-aware softplus prologue (softplus_b with eo_cols: the forward's hidden layers, a tenth of its
// work, stay on the kernel above): layer 0 of the forward, every weight- and data-gradient contraction.
constexpr int T2M = 64, T2N = 128;

__global__ void __launch_bounds__(256) gemm_generic2_kernel(NsvdGemm g) {
    __shared__ float As[2][TK][T2M + PAD];
    __shared__ float Bs[2][TK][T2N + PAD];
    const int bz = blockIdx.z;
    const float* A = g.A + (size_t)bz * g.bA;
    const float* Bm = g.B + (size_t)bz * g.bB;
    float* C = g.C + (size_t)bz * g.bC;
    const int m0 = blockIdx.y * T2M, n0 = blockIdx.x * T2N;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;  // this wave: rows 32 wm .., columns 64 wn .. (two 32-column blocks)
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    const bool a_kfast = (g.sAk == 1);
    const bool b_nfast = (g.sBn == 1);
    float ra[4], rb[8];
    // this thread's staging elements: (row / column, k) inside a tile - fixed for the whole loop
    int amm[4], akk[4], bnn[8], bkk[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (a_kfast) { akk[i] = t & 15; amm[i] = (t >> 4) + 16 * i; }
        else         { amm[i] = t & 63; akk[i] = (t >> 6) + 4 * i; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (b_nfast) { bnn[i] = t & 127; bkk[i] = (t >> 7) + 2 * i; }
        else         { bkk[i] = t & 15; bnn[i] = (t >> 4) + 16 * i; }
    }
    auto request = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = m0 + amm[i], gk = k0 + akk[i];
            ra[i] = (gm < g.M && gk < g.K) ? A[(size_t)gm * g.sAm + (size_t)gk * g.sAk] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gn = n0 + bnn[i], gk = k0 + bkk[i];
            rb[i] = (gn < g.N && gk < g.K) ? Bm[(size_t)gk * g.sBk + (size_t)gn * g.sBn] : 0.f;
        }
    };
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) As[buf][akk[i]][amm[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float v = rb[i];
            // (softplus of the zero padding would be log 2: padding stays zero)
            if (g.softplus_b && n0 + bnn[i] < g.N && k0 + bkk[i] < g.K) v = nsvd_softplus(v);
            Bs[buf][bkk[i]][bnn[i]] = v;
        }
    };
    request(0);
    stage(0, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += TK) {
        const bool more = k0 + TK < g.K;
        if (more) request(k0 + TK);
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            const float av = As[buf][kk + hi][32 * wm + li];
            const float b0 = Bs[buf][kk + hi][64 * wn + li], b1 = Bs[buf][kk + hi][64 * wn + 32 + li];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
        if (more) stage(buf ^ 1, k0 + TK);
        __syncthreads();  // tile t + 1 is staged, and every wave is done with tile t (the buffer tile t + 2 goes to)
        buf ^= 1;
    }
    const float* bias = g.bias ? g.bias + (size_t)bz * g.bBias : nullptr;
    const float* Z = g.Z ? g.Z + (size_t)bz * g.bZ : nullptr;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int gn = n0 + 64 * wn + 32 * blk + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (gm >= g.M || gn >= g.N) continue;
            const float bi = bias ? bias[gm] : 0.f;
            float v = (blk ? acc1[r] : acc0[r]) + ((g.eo_cols > 0 && gn >= g.eo_cols) ? 0.f : bi);
            if (g.sigmoid_mul) v *= nsvd_sigmoid(Z[(size_t)gm * g.sZm + gn]);
            C[(size_t)gm * g.sCm + gn] = v;
        }
    }
}

__global__ void __launch_bounds__(256) rowsum_kernel(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                     int n, long ld) {
    // one wave per row
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const float* p = in + (size_t)wave * ld;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += p[j];
    s = nsvd_wave_sum(s);
    if (lane == 0) out[wave] = s;
}

}  // namespace

int nsvd_gemm_generic(const NsvdGemm& g, hipStream_t s) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.batch <= 0) return NSVD_EINVAL;
    // the stencil-aware softplus prologue (three loads per staged element) stays on the 64 x 64 kernel; everything else takes
    // the pipelined 64 x 128 one (NSVD_GEMM_GENERIC2=0: the old kernel everywhere, for A/B measurements)
    static const char* e2 = getenv("NSVD_GEMM_GENERIC2");
    if (!(g.softplus_b && g.eo_cols > 0) && !(e2 && e2[0] == '0')) {
        dim3 grid2(nsvd_cdiv(g.N, T2N), nsvd_cdiv(g.M, T2M), g.batch);
        hipLaunchKernelGGL(gemm_generic2_kernel, grid2, dim3(256), 0, s, g);
        NSVD_CHECK_LAUNCH();
        return 0;
    }
    dim3 grid(nsvd_cdiv(g.N, TN), nsvd_cdiv(g.M, TM), g.batch);
    hipLaunchKernelGGL(gemm_generic_kernel, grid, dim3(256), 0, s, g);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_rowsum(const float* in, float* out, int rows, int n, long ld, hipStream_t s) {
    const int blocks = nsvd_cdiv(rows, 4);
    hipLaunchKernelGGL(rowsum_kernel, dim3(blocks), dim3(256), 0, s, in, out, rows, n, ld);
    NSVD_CHECK_LAUNCH();
    return 0;
}
