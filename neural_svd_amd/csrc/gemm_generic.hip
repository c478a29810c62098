// Generic strided, batched fp32 GEMM for shapes the fused MFMA kernels do not cover
// (hidden widths that are not 128, ragged batches). LDS-tiled 64 x 64 x 16; the products on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32: a wave owns one 32 x 32 quadrant of the tile, one MFMA and two 4-byte LDS reads per pair of
// k) since round 4 - rounds 1-3 multiplied on the vector ALU (4 x 4 register tiles, 36 TFLOP/s).
// Used for: ParallelMLP layers (reference mlp.py:204-221), their data gradients and weight
// gradients (what autograd derives for those einsums).
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
