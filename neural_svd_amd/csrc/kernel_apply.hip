// Dense kernel operator applied to a minibatch (the "Gram path" of the kernel-operator configuration):
//     Kf[i][l] = scale * sum_k K[rows_i][cols_k] f[k][l],        i < B1, k < B2, l < L
// K: (N, ldk) row-major symmetric PSD matrix on N points; rows / cols: minibatch indices (with replacement).
// The reference defines only the consumer (`get_approx_kernel_op(x)(model, x, importance) -> (Kf, f)`,
// methods/nestedlora.py:230-252) and ships no kernel operator, so this is the build's own definition
// (cfg4) - restated in float64 by oracle/nsvd_oracle.py:kernel_apply.
// Form used: scatter the batch into the index space, S^T[l][j] = sum_{k: cols_k = j} f[k][l], then
//     Kf = K[rows, :] S    -  a (B1 x N) . (N x L) contraction with GATHERED rows of K, both operands K-contiguous,
// on the fp32-input MFMA through the shared tile routine (tile_nt.h), split over N so that the grid fills the chip,
// partial tiles reduced in slice order (bit-reproducible when cols has no duplicates; duplicates are summed with
// float atomics in the scatter).
#include <string.h>
#include "nsvd_kernels.h"
#include "tile_nt.h"
#include "tile128_dma.h"

namespace {

constexpr int T = NSVD_TNT_T, KC = NSVD_TNT_KC;

struct KaWs {
    float* ST;      // (Lp, Np) scattered batch, zero padded
    float* part;    // (S, B1p, Lp) split-K partial tiles
    int Np, Lp, B1p, S;
    int dma;        // 1: 128-row tiles on the LDS-DMA ring (ka_gemm_dma_kernel), 0: 64-row tiles (ka_gemm_kernel)
    int all_rows;   // 1: the contraction runs over ALL N rows of K once and the output rows are gathered from it
    size_t bytes;
};

KaWs carve(void* base, int N, int B1, int L) {
    KaWs w;
    w.Np = nsvd_cdiv(N, KC) * KC;
    w.Lp = nsvd_cdiv(L, T) * T;
    // more output rows than points (head-sharded runs: the batch grows with the world size, the point set does not):
    // rows drawn with replacement repeat, and the gathered rows of K are what the kernel streams from HBM - multiply
    // every row of K once (N rows) and gather the B1 output rows from the product
    w.all_rows = B1 > N ? 1 : 0;
    w.B1p = nsvd_cdiv(w.all_rows ? N : B1, T) * T;
    // 128 gathered rows per tile through the LDS-DMA ring of tile128_dma.h where the row count allows (the gathered rows
    // of K are what this kernel streams from HBM: 328 MB at configs[3]; the ring keeps two chunks of every row in flight)
    w.dma = (w.B1p % 128 == 0 && w.B1p >= 1024) ? 1 : 0;
    int S = 1;
    if (w.dma) {
        const int tiles = (w.B1p / 128) * (w.Lp / T), chunks32 = w.Np / 32;
        while (tiles * S < 512 && chunks32 / (2 * S) >= 16) S *= 2;
    } else {
        const int tiles = (w.B1p / T) * (w.Lp / T), chunks = w.Np / KC;
        while (tiles * S < 512 && chunks / (2 * S) >= 8) S *= 2;  // enough blocks for two per CU, >= 8 chunks per slice
    }
    w.S = S;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { float* r = (float*)(p + off); off += nsvd_align(n * sizeof(float)); return r; };
    w.ST = take((size_t)w.Lp * w.Np);
    w.part = take((size_t)S * w.B1p * w.Lp);
    w.bytes = off;
    return w;
}

__global__ void __launch_bounds__(256) ka_zero_kernel(float4* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void __launch_bounds__(256) ka_scatter_kernel(const float* __restrict__ f, const long long* __restrict__ cols,
                                                         int B2, int L, int N, float* __restrict__ ST, int Np) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B2 * L) return;
    const int k = i / L, l = i - k * L;
    const long long j = cols[k];
    if (j < 0 || j >= N) return;  // out-of-range indices contribute nothing
    atomicAdd(ST + (size_t)l * Np + j, f[i]);
}

// tile (tb, tl) x slice: rows of K gathered by `rows`, contraction range [c0, c1) chunks of the point index
__global__ void __launch_bounds__(256, 2) ka_gemm_kernel(const float* __restrict__ K, long ldk, int N,
                                                         const long long* __restrict__ rows, int B1, KaWs w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tb = blockIdx.x, tl = blockIdx.y, slice = blockIdx.z;
    const int chunks = w.Np / KC;
    const int c0 = (int)((long)chunks * slice / w.S), c1 = (int)((long)chunks * (slice + 1) / w.S);
    const int t = threadIdx.x, lr0 = t >> 4, lc = (t & 15) * 4;
    const float* ap[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = tb * T + lr0 + 16 * i;
        // padding rows of the tile read row 0; their results are never used. rows == nullptr: row r of K itself
        long long g = r < B1 ? (rows ? rows[r] : (long long)r) : 0;
        if (g < 0 || g >= N) g = 0;
        ap[i] = K + g * ldk + lc + (long)c0 * KC;
    }
    nsvd_f32x16 acc = {0};
    float unused[4] = {0.f, 0.f, 0.f, 0.f};
    nsvd_tile_nt_rows<false>(ap[0], ap[1], ap[2], ap[3], w.ST + ((size_t)tl * T + lr0) * w.Np + lc + (size_t)c0 * KC,
                             w.Np, c1 - c0, lds, acc, unused);
    const int lane = t & 63, wv = t >> 6;
    float* out = w.part + (size_t)slice * w.B1p * w.Lp;
    const int col = tl * T + (wv >> 1) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = tb * T + (wv & 1) * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
        out[(size_t)row * w.Lp + col] = acc[r];
    }
}

// the same contraction on 128 x 64 tiles with both operands by LDS-DMA (tile128_dma.h, gathered A rows): no staging
// registers, four half chunks of every row in flight
__global__ void __launch_bounds__(256, 2) ka_gemm_dma_kernel(const float* __restrict__ K, long ldk, int N,
                                                             const long long* __restrict__ rows, int B1, KaWs w) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    using namespace nsvd_pmlp;
    const int tb = blockIdx.x, tl = blockIdx.y, slice = blockIdx.z;
    const int chunks = w.Np / 32;
    const int c0 = (int)((long)chunks * slice / w.S), c1 = (int)((long)chunks * (slice + 1) / w.S);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned ga[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = tb * 128 + 32 * wv + (lane >> 2) + 16 * i;
        long long g = r < B1 ? (rows ? rows[r] : (long long)r) : 0;  // (padding rows read row 0; never used)
        if (g < 0 || g >= N) g = 0;
        ga[i] = (unsigned)((size_t)g * (size_t)ldk * sizeof(float));
    }
    f32x16 acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    Tile128NoHook none;
    nsvd_tile128_dma<Tile128NoHook, 1, false, true>(K + (size_t)c0 * 32, w.ST + (size_t)tl * T * w.Np + (size_t)c0 * 32, 0u,
                                                    (unsigned)w.Np, c1 - c0, lds, acc, none, ga[0], ga[1]);
    // wave (wm, wn): rows 64 wm + 32 i of the tile, columns 32 wn of the 64
    const int li = lane & 31, hi = lane >> 5, wm = wv >> 1, wn = wv & 1;
    float* out = w.part + (size_t)slice * w.B1p * w.Lp;
    const int col = tl * T + 32 * wn + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tb * 128 + 64 * wm + 32 * i + acc_row(r, hi);
            out[(size_t)row * w.Lp + col] = acc[i][0][r];
        }
}

__global__ void __launch_bounds__(256) ka_reduce_kernel(KaWs w, const long long* __restrict__ rows, int N, int B1, int L,
                                                        float scale, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B1 * L) return;
    const int b = i / L, l = i - b * L;
    const long long g = rows[b];
    if (g < 0 || g >= N) {
        out[i] = 0.f;
        return;
    }
    const size_t pr = w.all_rows ? (size_t)g : (size_t)b;  // product row: of K's row g, or of this output row
    float s = 0.f;
    for (int sl = 0; sl < w.S; ++sl) s += w.part[((size_t)sl * w.B1p + pr) * w.Lp + l];
    out[i] = scale * s;
}

size_t ka_lds_bytes() {
    static const size_t bytes = [] {
        const size_t b = NSVD_TNT_FLOATS * sizeof(float);
        (void)hipFuncSetAttribute((const void*)ka_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
        return b;
    }();
    return bytes;
}

}  // namespace

extern "C" size_t nsvd_kernel_apply_workspace_bytes(int N, int B1, int L) {
    if (N <= 0 || B1 <= 0 || L <= 0) return 0;
    return carve(nullptr, N, B1, L).bytes;
}

extern "C" int nsvd_kernel_apply(const float* K, size_t ldk, int N, const long long* rows, int B1,
                                 const long long* cols, int B2, const float* f, int L, float scale, float* out,
                                 void* ws, size_t ws_bytes, void* stream) {
    if (!K || !rows || !cols || !f || !out || !ws || N <= 0 || B1 <= 0 || B2 <= 0 || L <= 0) return NSVD_EINVAL;
    const KaWs w = carve(ws, N, B1, L);
    // the tile routine reads whole 64-float chunks of a row: the leading dimension must cover the padded width
    if (ldk < (size_t)w.Np || (ldk & 3) != 0 || ((uintptr_t)K & 15) != 0) return NSVD_EINVAL;
    if (ws_bytes < w.bytes || ((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const size_t n4 = (size_t)w.Lp * w.Np / 4;
    ka_zero_kernel<<<(unsigned)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256), 256, 0, s>>>((float4*)w.ST, n4);
    NSVD_CHECK_LAUNCH();
    ka_scatter_kernel<<<nsvd_cdiv(B2 * L, 256), 256, 0, s>>>(f, cols, B2, L, N, w.ST, w.Np);
    NSVD_CHECK_LAUNCH();
    nsvd_prof_begin(s);  // bench.py --config cfg4 brackets the contraction (nsvd_profile_next_forward)
    if (w.dma && (size_t)N * ldk * sizeof(float) < ((size_t)1 << 32)) {  // (32-bit row offsets)
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)ka_gemm_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(nsvd_pmlp::T128D_LDS_FLOATS * sizeof(float)));
            attr_set = true;
        }
        ka_gemm_dma_kernel<<<dim3(w.B1p / 128, w.Lp / T, w.S), 256, nsvd_pmlp::T128D_LDS_FLOATS * sizeof(float), s>>>(
            K, (long)ldk, N, w.all_rows ? nullptr : rows, w.all_rows ? N : B1, w);
    } else {
        ka_gemm_kernel<<<dim3(w.B1p / T, w.Lp / T, w.S), 256, ka_lds_bytes(), s>>>(K, (long)ldk, N,
                                                                                    w.all_rows ? nullptr : rows,
                                                                                    w.all_rows ? N : B1, w);
    }
    nsvd_prof_end(s);
    NSVD_CHECK_LAUNCH();
    ka_reduce_kernel<<<nsvd_cdiv(B1 * L, 256), 256, 0, s>>>(w, rows, N, B1, L, scale, out);
    NSVD_CHECK_LAUNCH();
    return 0;
}
