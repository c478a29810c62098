// NestedLoRA loss for canonical dependence kernels (the two-tower / cross-domain path).
//   reference: NestedLoRALossFunctionForCDK.forward   methods/nestedlora.py:273-306
//              NestedLoRALossFunctionForCDK.backward  methods/nestedlora.py:309-332
//              off_diagonal                           methods/utils.py:16-22
// With f~ = bw * [1, f], g~ = bw * [1, g] (B x Lp, Lp = L + first, first = set_first_mode_const):
//   lam_f = f~^T f~ / B, lam_g = g~^T g~ / B                       (Lp x Lp, contraction over the batch)
//   loss  = -2 mean_b sum_l v_l f~ g~ + sum(M * lam_f * lam_g)
//   gram  = f~ g~^T -> rs_joint = diag, rs_indep = off-diagonal    (B x B, contraction over the modes)
//   d f   = -(2/B) g~ v + (2/B) f~ (M * lam_g)   (constant column dropped), same for g with f <-> g.
// At the reference's configuration (L = 512, B = 1024) these are five 0.5-1 GFLOP contractions: they run
// on the fp32-input MFMA (v_mfma_f32_32x32x2_f32), all through ONE tile routine for C = A B^T with both
// operands K-contiguous.  A staging kernel writes the padded, weighted features once in both orientations
// (zero-filled, so the contraction loops carry no bounds checks); the loss scalars are reduced from per-block
// partial sums in a fixed order (no float atomics: bit-reproducible).
//
// Tiling is chosen against wave quantisation (256 CUs, one 64 x 64 tile keeps a CU's four MFMA pipes busy):
// the 64-tiles of the mode axis start at the first FEATURE mode, so L = 512 is 8 tiles, not 9 with one live
// column; the constant mode's row of lam is a matrix-vector product done by a few plain-FMA blocks
// ("border"). Backward = 16 x 8 x 2 = 256 tiles = one round; gram = 256 tiles; lam = 2 x 36 upper-triangular
// tiles (lam is symmetric) x split-K 2.
//
// Launches:  cdk_stage (pad + transpose + operator-term partials)
//            cdk_forward_gemm (border rows, gram tiles, lam_f / lam_g tiles in one grid)
//            cdk_finish (M * lam, transposed for the backward; metric-term partials)
//            cdk_loss_reduce (one block: loss[3] from the partials; the fused training step lets its optimiser kernel
//                             do this sum instead: nsvd_cdk_loss_forward_parts)
//            cdk_backward_gemm (both gradients in one grid)
#include "nsvd_kernels.h"
#include "tile_nt.h"

namespace {

typedef nsvd_f32x16 floatx16;

constexpr int T = NSVD_TNT_T;    // tile edge (rows of A x rows of B per block of 4 waves)
constexpr int TS = 32;           // tile edge of the staging / finish kernels
constexpr int KC = NSVD_TNT_KC;  // contraction chunk
constexpr int TILE_FLOATS = NSVD_TNT_FLOATS;  // double-buffered A and B chunks: 68 KB
constexpr int NB0 = 8;           // batch slices of the constant-mode row (border blocks)
constexpr int SPLIT_CHUNKS = 8;  // batch contraction: about this many chunks per block (split-K)

// Mode index i in [0, Lp): i = 0 is the constant mode when first = 1, feature l sits at i = first + l.
struct CdkWs {
    float *ft, *gt;        // (Bp, LD) padded weighted features, K-contiguous for gram / backward
    float *fT, *gT;        // (LD, Bp) transposes, K-contiguous for the batch contraction
    float *lam_f, *lam_g;  // (nsplit, LD, LD) unscaled split-K partials of the feature x feature block; only the
                           // 64-tiles on or above the diagonal are written (lam is symmetric)
    float *lam0_f, *lam0_g;  // (NB0, LD) unscaled partials of the constant-mode row sum_b f~[b][0] f~[b][j]
    float *MlfT, *MlgT;    // (LD, LD): [j][i] = M[i][j] * lam[i][j], rows j in [first, first + R)
    float *part_op, *part_met;
    int Bp, LD, Lp, R, first, nstage, nmain, nfin, nsplit;
    size_t bytes;
};

__host__ CdkWs carve(void* base, int B, int L, int first) {
    CdkWs w;
    w.first = first ? 1 : 0;
    w.Lp = L + w.first;
    w.Bp = nsvd_cdiv(B, T) * T;
    w.R = nsvd_cdiv(L, T) * T;                // feature modes, padded: the tiled range [first, first + R)
    w.LD = nsvd_cdiv(w.R + w.first, T) * T;   // leading dimension / contraction length over the modes
    w.nstage = (w.Bp / TS) * (w.LD / TS);
    w.nmain = (w.R / TS) * (w.R / TS);
    w.nfin = w.nmain + (w.first ? nsvd_cdiv(w.LD, 256) : 0);
    w.nsplit = nsvd_cdiv(w.Bp / KC, SPLIT_CHUNKS);
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t n) { float* r = (float*)(p + off); off += nsvd_align(n * sizeof(float)); return r; };
    const size_t feat = (size_t)w.Bp * w.LD, sq = (size_t)w.LD * w.LD;
    w.ft = take(feat); w.gt = take(feat); w.fT = take(feat); w.gT = take(feat);
    w.lam_f = take(sq * w.nsplit); w.lam_g = take(sq * w.nsplit);
    w.lam0_f = take((size_t)NB0 * w.LD); w.lam0_g = take((size_t)NB0 * w.LD);
    w.MlfT = take(sq); w.MlgT = take(sq);
    w.part_op = take(w.nstage); w.part_met = take(w.nfin);
    w.bytes = off;
    return w;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = nsvd_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------------------ staging
// One 32 x 32 tile of f~ and g~ per block (4 elements per thread, everything in flight at once): pad the
// constant mode, apply the row weights, zero-fill, store row-major and (through LDS) transposed; partial of
// sum_b sum_l v_l f~ g~.
__global__ void __launch_bounds__(256) cdk_stage_kernel(const float* __restrict__ f, const float* __restrict__ g,
                                                        const float* __restrict__ bw, const float* __restrict__ v,
                                                        int B, int L, CdkWs w) {
    __shared__ float tf[TS][TS + 1], tg[TS][TS + 1];
    __shared__ float red[4];
    const int tb = blockIdx.x, tl = blockIdx.y, first = w.first;
    const int c = threadIdx.x & 31, r8 = threadIdx.x >> 5;
    const int j = tl * TS + c;
    const bool jin = j < w.Lp, jconst = first && j == 0;
    const float vj = jin ? v[j] : 0.f;
    float fv[4], gv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int b = tb * TS + r8 + 8 * q;
        fv[q] = gv[q] = 0.f;
        if (b < B && jin) {
            const float wb = bw ? bw[b] : 1.f;
            fv[q] = wb * (jconst ? 1.f : f[(size_t)b * L + (j - first)]);
            gv[q] = wb * (jconst ? 1.f : g[(size_t)b * L + (j - first)]);
        }
    }
    float op = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = r8 + 8 * q;
        const size_t o = (size_t)(tb * TS + r) * w.LD + j;
        w.ft[o] = fv[q];
        w.gt[o] = gv[q];
        tf[r][c] = fv[q];
        tg[r][c] = gv[q];
        op = fmaf(vj * fv[q], gv[q], op);
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int jr = r8 + 8 * q;  // mode row of the transposed tile; c = batch column
        const size_t o = (size_t)(tl * TS + jr) * w.Bp + tb * TS + c;
        w.fT[o] = tf[c][jr];
        w.gT[o] = tg[c][jr];
    }
    const float s = block_sum_256(op, red);
    if (threadIdx.x == 0) w.part_op[blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// ------------------------------------------------------------------------------------------ tile GEMM
// tile_nt.h: acc(32x32 of this wave) += A[64 rows][k0 : k1] . B[64 rows][k0 : k1]^T
__device__ __forceinline__ void tile_nt(const float* __restrict__ A, long lda, const float* __restrict__ Bm,
                                        long ldb, int k0, int k1, float* __restrict__ lds, floatx16& acc) {
    float unused[4] = {0.f, 0.f, 0.f, 0.f};
    nsvd_tile_nt<false>(A, lda, Bm, ldb, k0, k1, lds, acc, unused);
}

// accumulator element r of this lane sits at (row, col) of the wave's 32 x 32 block
__device__ __forceinline__ int acc_row(int lane, int r) { return (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3); }

// ------------------------------------------------------------------------------------------ forward GEMMs
// One grid: [border blocks | gram tiles | lam tiles].
//   border (first = 1): 64 modes x 1/NB0 of the batch per block, partials of lam0[j] = sum_b f~[b][0] f~[b][j]
//     by plain FMAs (latency-bound, scheduled first so they hide under the tiles);
//   gram: nb x nb tiles, contraction over the LD modes;
//   lam_f, lam_g: 64-tiles (ti <= tj) of the feature block x nsplit slices of the batch.
__global__ void __launch_bounds__(256) cdk_forward_gemm_kernel(CdkWs w, int B, int nbord, int ngram,
                                                                  float* __restrict__ rs_joint,
                                                                  float* __restrict__ rs_indep) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nr = w.R / T, nb = w.Bp / T;
    int bid = blockIdx.x;
    if (bid < nbord) {
        const int slice = bid % NB0, j = (bid / NB0) * T + lane, q = slice * 4 + wv, rows = w.Bp / (4 * NB0);
        const float* pf = w.ft + (size_t)q * rows * w.LD;
        const float* pg = w.gt + (size_t)q * rows * w.LD;
        float sf = 0.f, sg = 0.f;
#pragma unroll 8
        for (int b = 0; b < rows; ++b) {
            sf = fmaf(pf[(size_t)b * w.LD], pf[(size_t)b * w.LD + j], sf);
            sg = fmaf(pg[(size_t)b * w.LD], pg[(size_t)b * w.LD + j], sg);
        }
        lds[wv * T + lane] = sf;
        lds[4 * T + wv * T + lane] = sg;
        __syncthreads();
        if (wv == 0) {
            w.lam0_f[slice * w.LD + j] = (lds[lane] + lds[T + lane]) + (lds[2 * T + lane] + lds[3 * T + lane]);
            w.lam0_g[slice * w.LD + j] =
                (lds[4 * T + lane] + lds[5 * T + lane]) + (lds[6 * T + lane] + lds[7 * T + lane]);
        }
        return;
    }
    bid -= nbord;
    floatx16 acc = {0};
    if (bid < ngram) {
        const int ta = bid / nb, tb = bid - ta * nb;
        tile_nt(w.ft + (size_t)ta * T * w.LD, w.LD, w.gt + (size_t)tb * T * w.LD, w.LD, 0, w.LD, lds, acc);
        const int b2 = tb * T + (wv >> 1) * 32 + (lane & 31);
        if (b2 >= B) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b1 = ta * T + (wv & 1) * 32 + acc_row(lane, r);
            if (b1 >= B) continue;
            if (b1 == b2) {
                if (rs_joint) rs_joint[b1] = acc[r];
            } else if (rs_indep) {
                // off_diagonal (view (B-1, B+1) of the flat gram minus its first column) = row-major order
                // of the gram with the diagonal removed
                rs_indep[(size_t)b1 * (B - 1) + b2 - (b2 > b1 ? 1 : 0)] = acc[r];
            }
        }
        return;
    }
    bid -= ngram;
    const int ntri = nr * (nr + 1) / 2;
    const int split = bid / (2 * ntri);
    int rem = bid - split * 2 * ntri;
    const int which = rem / ntri;
    rem -= which * ntri;
    int ti = 0;
    while (rem >= nr - ti) { rem -= nr - ti; ++ti; }  // row ti of the triangle holds nr - ti tiles
    const int tj = ti + rem;
    const int chunks = w.Bp / KC;
    const int c0 = (int)((long)chunks * split / w.nsplit), c1 = (int)((long)chunks * (split + 1) / w.nsplit);
    const float* X = (which ? w.gT : w.fT) + (size_t)w.first * w.Bp;
    tile_nt(X + (size_t)ti * T * w.Bp, w.Bp, X + (size_t)tj * T * w.Bp, w.Bp, c0 * KC, c1 * KC, lds, acc);
    float* out = (which ? w.lam_g : w.lam_f) + (size_t)split * w.LD * w.LD + (size_t)w.first * (w.LD + 1);
    const int col = tj * T + (wv >> 1) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ti * T + (wv & 1) * 32 + acc_row(lane, r);
        out[(size_t)row * w.LD + col] = acc[r];
    }
}

// ------------------------------------------------------------------------------------------ finish
// Blocks [0, nmain): 32 x 32 tile (ti, tj) of the feature block: lam = (sum of the split-K partials) / B,
// mirrored from the stored triangle of 64-tiles when it lies below it; M * lam_f, M * lam_g stored transposed
// (the backward contracts over i); partial of sum M lam_f lam_g.
// The last row of tiles also zero-fills the contraction padding of its rows.
// Remaining blocks (first = 1): one mode j per thread: the constant mode's column of M * lam and the constant
// row / column terms of the metric sum.
__global__ void __launch_bounds__(256) cdk_finish_kernel(CdkWs w, const float* __restrict__ M, int B) {
    __shared__ float tf[TS][TS + 1], tg[TS][TS + 1];
    __shared__ float red[4];
    const float inv = 1.0f / (float)B;
    const int first = w.first;
    float met = 0.f;
    if ((int)blockIdx.x >= w.nmain) {
        const int j = ((int)blockIdx.x - w.nmain) * 256 + threadIdx.x;
        if (j < w.LD) {
            float lf = 0.f, lg = 0.f;
#pragma unroll
            for (int s = 0; s < NB0; ++s) {
                lf += w.lam0_f[s * w.LD + j];
                lg += w.lam0_g[s * w.LD + j];
            }
            lf *= inv;
            lg *= inv;
            if (j < w.Lp) {
                const float m = j == 0 ? M[0] : M[j] + M[(size_t)j * w.Lp];  // M[0][j] + M[j][0]
                met = m * lf * lg;
            }
            if (j >= first && j < first + w.R) {
                const float m0 = j < w.Lp ? M[j] : 0.f;  // M[i = 0][j]
                w.MlfT[(size_t)j * w.LD] = m0 * lf;
                w.MlgT[(size_t)j * w.LD] = m0 * lg;
            }
        }
    } else {
        const int nt = w.R / TS;
        const int ti = blockIdx.x / nt, tj = blockIdx.x - ti * nt;
        const int c = threadIdx.x & 31, r8 = threadIdx.x >> 5;
        const bool upper = (ti >> 1) <= (tj >> 1);
        const int sa = upper ? ti : tj, sb = upper ? tj : ti;  // stored 32-tile
        const size_t sq = (size_t)w.LD * w.LD;
        float lf[4] = {0.f, 0.f, 0.f, 0.f}, lg[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < w.nsplit; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t o = s * sq + (size_t)(first + sa * TS + r8 + 8 * q) * w.LD + first + sb * TS + c;
                lf[q] += w.lam_f[o];
                lg[q] += w.lam_g[o];
            }
        }
        if (!upper) {  // block-uniform: transpose the stored tile through LDS
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                tf[r8 + 8 * q][c] = lf[q];
                tg[r8 + 8 * q][c] = lg[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lf[q] = tf[c][r8 + 8 * q];
                lg[q] = tg[c][r8 + 8 * q];
            }
            __syncthreads();
        }
        const int j = first + tj * TS + c;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = r8 + 8 * q, i = first + ti * TS + r;
            float mf = 0.f, mg = 0.f;
            if (i < w.Lp && j < w.Lp) {
                const float m = M[(size_t)i * w.Lp + j];
                mf = m * (lf[q] * inv);
                mg = m * (lg[q] * inv);
                met = fmaf(mf, lg[q] * inv, met);
            }
            tf[r][c] = mf;
            tg[r][c] = mg;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jr = r8 + 8 * q;
            const size_t o = (size_t)(first + tj * TS + jr) * w.LD + first + ti * TS + c;
            w.MlfT[o] = tf[c][jr];
            w.MlgT[o] = tg[c][jr];
        }
        if (first && ti == nt - 1) {  // zero the contraction padding [first + R, LD) of this tile's rows
            for (int e = threadIdx.x; e < TS * T; e += 256) {
                const int i = first + w.R + (e & (T - 1));
                if (i < w.LD) {
                    const size_t o = (size_t)(first + tj * TS + (e >> 6)) * w.LD + i;
                    w.MlfT[o] = 0.f;
                    w.MlgT[o] = 0.f;
                }
            }
        }
    }
    const float s = block_sum_256(met, red);
    if (threadIdx.x == 0) w.part_met[blockIdx.x] = s;
}

// loss[0..2] from the per-block partials, added in index order. A separate one-block launch: the
// "last block reduces" idiom needs a device-scope release per block, which on this multi-XCD part is an L2
// writeback per block (measured: 17 us for the finish kernel with it, 5 without).
__global__ void __launch_bounds__(256) cdk_loss_reduce_kernel(NsvdCdkLossParts lp, float* __restrict__ loss) {
    __shared__ float red[8];
    nsvd_cdk_loss_sum(lp, red, loss);
}

// ------------------------------------------------------------------------------------------ backward
// Tile (tb, tj) of the (B, L) gradient; blockIdx.z = 0: grad_f = f~ (M lam_g), 1: grad_g = g~ (M lam_f). The
// epilogue subtracts the v term and scales by (2/B) * grad_out; the constant column is never computed.
__global__ void __launch_bounds__(256) cdk_backward_gemm_kernel(CdkWs w, const float* __restrict__ v, int B,
                                                                   int L, const float* __restrict__ grad_out,
                                                                   float* __restrict__ grad_f,
                                                                   float* __restrict__ grad_g, int only) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tb = blockIdx.x, tj = blockIdx.y, which = only >= 0 ? only : (int)blockIdx.z;
    const float* X = which ? w.gt : w.ft;     // contracted operand
    const float* Y = which ? w.ft : w.gt;     // the other tower: the -(2/B) v y term
    const float* Mt = (which ? w.MlfT : w.MlgT) + (size_t)w.first * w.LD;
    float* out = which ? grad_g : grad_f;
    floatx16 acc = {0};
    tile_nt(X + (size_t)tb * T * w.LD, w.LD, Mt + (size_t)tj * T * w.LD, w.LD, 0, w.LD, lds, acc);
    const int l = tj * T + (wv >> 1) * 32 + (lane & 31);  // feature index; mode index j = first + l
    const int j = w.first + l;
    float y[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)  // rows up to Bp exist in the padded buffer: all 16 loads in flight at once
        y[r] = Y[(size_t)(tb * T + (wv & 1) * 32 + acc_row(lane, r)) * w.LD + j];
    if (l >= L) return;
    const float sc = (grad_out ? *grad_out : 1.0f) * (2.0f / (float)B);
    const float vj = v[j];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int b = tb * T + (wv & 1) * 32 + acc_row(lane, r);
        if (b < B) out[(size_t)b * L + l] = sc * (acc[r] - y[r] * vj);
    }
}

// double-buffered 64-wide chunks: 68 KB, above the 64 KB default -> opt in once per process
size_t gemm_lds_bytes() {
    static const size_t bytes = [] {
        const size_t b = TILE_FLOATS * sizeof(float);
        (void)hipFuncSetAttribute((const void*)cdk_forward_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)b);
        (void)hipFuncSetAttribute((const void*)cdk_backward_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)b);
        return b;
    }();
    return bytes;
}

}  // namespace

// ------------------------------------------------------------------------------------------ C ABI
extern "C" size_t nsvd_cdk_workspace_bytes(int B, int L, int set_first_mode_const) {
    if (B <= 0 || L <= 0) return 0;
    return carve(nullptr, B, L, set_first_mode_const).bytes;
}

int nsvd_cdk_loss_forward_parts(const float* f, const float* g, const float* batch_weights, const float* v,
                                const float* M, int B, int L, int set_first_mode_const, float* rs_joint,
                                float* rs_indep, void* ws, size_t ws_bytes, NsvdCdkLossParts* parts, hipStream_t s) {
    if (!f || !g || !v || !M || !ws || !parts || B <= 0 || L <= 0) return NSVD_EINVAL;
    if (rs_indep && B < 2) return NSVD_EINVAL;
    const CdkWs w = carve(ws, B, L, set_first_mode_const);
    if (ws_bytes < w.bytes) return NSVD_EINVAL;
    const int nr = w.R / T, nb = w.Bp / T;
    cdk_stage_kernel<<<dim3(w.Bp / TS, w.LD / TS), 256, 0, s>>>(f, g, batch_weights, v, B, L, w);
    NSVD_CHECK_LAUNCH();
    const int nbord = w.first ? NB0 * (w.LD / T) : 0;
    const int ngram = (rs_joint || rs_indep) ? nb * nb : 0;
    const int nlam = 2 * (nr * (nr + 1) / 2) * w.nsplit;
    cdk_forward_gemm_kernel<<<nbord + ngram + nlam, 256, gemm_lds_bytes(), s>>>(w, B, nbord, ngram, rs_joint,
                                                                                 rs_indep);
    NSVD_CHECK_LAUNCH();
    cdk_finish_kernel<<<w.nfin, 256, 0, s>>>(w, M, B);
    NSVD_CHECK_LAUNCH();
    parts->part_op = w.part_op; parts->part_met = w.part_met;
    parts->nstage = w.nstage; parts->nfin = w.nfin; parts->B = B;
    return 0;
}

extern "C" int nsvd_cdk_loss_forward(const float* f, const float* g, const float* batch_weights, const float* v,
                                     const float* M, int B, int L, int set_first_mode_const, float* loss,
                                     float* rs_joint, float* rs_indep, void* ws, size_t ws_bytes, void* stream) {
    if (!loss) return NSVD_EINVAL;
    NsvdCdkLossParts lp;
    hipStream_t s = (hipStream_t)stream;
    const int rc = nsvd_cdk_loss_forward_parts(f, g, batch_weights, v, M, B, L, set_first_mode_const, rs_joint, rs_indep,
                                               ws, ws_bytes, &lp, s);
    if (rc) return rc;
    cdk_loss_reduce_kernel<<<1, 256, 0, s>>>(lp, loss);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_cdk_loss_backward(const float* v, int B, int L, int set_first_mode_const, const float* grad_out,
                                      float* grad_f, float* grad_g, void* ws, size_t ws_bytes, void* stream) {
    if (!v || !ws || B <= 0 || L <= 0 || (!grad_f && !grad_g)) return NSVD_EINVAL;
    const CdkWs w = carve(ws, B, L, set_first_mode_const);
    if (ws_bytes < w.bytes) return NSVD_EINVAL;
    const int both = grad_f && grad_g;
    cdk_backward_gemm_kernel<<<dim3(w.Bp / T, w.R / T, both ? 2 : 1), 256, gemm_lds_bytes(), (hipStream_t)stream>>>(
        w, v, B, L, grad_out, grad_f, grad_g, both ? -1 : (grad_g ? 1 : 0));
    NSVD_CHECK_LAUNCH();
    return 0;
}
