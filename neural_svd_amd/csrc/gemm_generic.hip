// Generic strided, batched fp32 GEMM for shapes the fused MFMA kernels do not cover (hidden widths that are not 128,
// ragged batches, D > 3), and the in-place activation pass between two layers. Products on the fp32 MFMA
// (v_mfma_f32_32x32x2_f32) since round 4 - rounds 1-3 multiplied on the vector ALU (4 x 4 register tiles, 36 TFLOP/s).
// Used for: ParallelMLP layers (reference mlp.py:204-221), their data gradients and weight gradients (what autograd
// derives for those einsums). Round 6: the vectorised kernel (gemm_generic3_kernel) takes every aligned launch, the
// scalar one (gemm_generic2_kernel) the rest; the softplus is no longer a prologue of the NEXT layer's contraction
// (recomputed by every row tile: the hidden layers ran at 0.34 of peak) but ONE element-wise pass that turns a layer's
// pre-activations into activations in place (nsvd_softplus_inplace) - the backward reads the stored activations.
#include <stdint.h>
#include <stdlib.h>
#include "nsvd_kernels.h"

namespace {

constexpr int TK = 16, PAD = 4;

// The scalar kernel (any strides, any alignment): 64 x 128 x 16 LDS tiles, a wave owns 32 rows x 64 columns (two 32 x 32
// accumulators fed by ONE A value: three 4-byte LDS reads per two MFMAs), the LDS tiles double buffered (one workgroup
// barrier per 16 k), the staging values of tile t + 1 requested before tile t is multiplied and written behind it.
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
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) As[buf][akk[i]][amm[i]] = ra[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) Bs[buf][bkk[i]][bnn[i]] = rb[i];
    };
    request(0);
    stage(0);
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
        if (more) stage(buf ^ 1);
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
            if (g.sigmoid_mul) v *= nsvd_sigmoid_from_softplus(Z[(size_t)gm * g.sZm + gn]);
            C[(size_t)gm * g.sCm + gn] = v;
        }
    }
}

// Round 6, the vectorised form: the same contraction for launches whose fast axes are 16-byte aligned (every launch of
// the generic path at batch sizes that are multiples of 4). What changes against the kernel above:
//  * 16-byte global loads along each operand's contiguous axis (k-contiguous operands are transposed on their way into
//    the LDS: four 4-byte writes, conflict-free with the 4-float row padding; m / n-contiguous ones go in as ds_write_b128);
//  * no branch around any load: ragged M / N edges CLAMP the row / column (the values land in accumulator rows / columns
//    that are never stored), so a K step's loads issue back to back and stay in flight under the previous step's MFMAs;
//  * larger tiles, chosen per launch: a wave owns MI x NI accumulators of 32 x 32 (2 x 2 waves): 128 x 128 (32 FLOP per
//    staged byte against 21 of the 64 x 128 tile), 64 x 256, 64 x 128 or 64 x 64 - the largest that still gives the chip
//    >= 1024 workgroups (four per CU: what the registers and the LDS let reside);
//  * K tails by select (an item past K loads from the operand's base and is staged as zero);
//  * the epilogue's loads (bias, the stored activation of the sigmoid factor) batched per accumulator block.
//  * KG > 1 (64 x 64 tiles only): a launch with too few workgroups for the chip (a hidden layer's weight gradient at
//    H = 64: 16 tiles, K = 512 - 32 serial K steps of one wave per SIMD, 19 us) gives each workgroup KG groups of four
//    waves; group kg multiplies the kg-th K range into its own accumulators (its own LDS stages), the partial tiles
//    meet in the LDS and are added in group order (fixed order: bit-reproducible), group 0 runs the epilogue.
template <int MI, int NI, bool AK, bool BK, int KG = 1>
__global__ void __launch_bounds__(256 * KG, KG > 1 ? 1 : (MI * NI >= 4 ? 3 : 4)) gemm_generic3_kernel(NsvdGemm g) {
    static_assert(KG == 1 || (MI == 1 && NI == 1), "the K groups exist for the 64 x 64 tile only");
    constexpr int TM3 = 64 * MI, TN3 = 64 * NI, LDA = TM3 + PAD, LDB = TN3 + PAD;
    __shared__ __attribute__((aligned(16))) float AsG[KG][2][TK][LDA];
    __shared__ __attribute__((aligned(16))) float BsG[KG][2][TK][LDB];
    const int kg = KG > 1 ? (int)(threadIdx.x >> 8) : 0;
    float (*As)[TK][LDA] = AsG[kg];
    float (*Bs)[TK][LDB] = BsG[kg];
    // this group's K range: whole K steps, the same trip count for every group (the barriers are the workgroup's)
    const int ksteps = ((g.K + TK - 1) / TK + KG - 1) / KG;
    const int kbeg = kg * ksteps * TK, kend = min(g.K, kbeg + ksteps * TK);
    const int bz = blockIdx.z;
    const float* A = g.A + (size_t)bz * g.bA;
    const float* Bm = g.B + (size_t)bz * g.bB;
    float* C = g.C + (size_t)bz * g.bC;
    const int m0 = blockIdx.y * TM3, n0 = blockIdx.x * TN3;
    const int t = threadIdx.x & 255;
    const int lane = t & 63, wv = t >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;  // this wave: rows 32 MI wm .., columns 32 NI wn ..
    typedef float f32x16_t __attribute__((ext_vector_type(16)));
    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging items of this thread (fixed for the whole loop): MI float4 of A, NI float4 of B
    const float* pa[MI];
    int wa[MI];  // LDS float offset inside a stage of As
    int ka[MI], kb[NI];  // first k of the item inside a K step
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int idx = t + 256 * i;
        if (AK) {  // A[m][k], k contiguous: a float4 = four k of one row
            const int q = idx & 3, m = idx >> 2;
            const int gm = min(m0 + m, g.M - 1);
            pa[i] = A + (size_t)gm * g.sAm + 4 * q + kbeg;
            wa[i] = 4 * q * LDA + m;
            ka[i] = 4 * q;
        } else {   // A[k][m], m contiguous: a float4 = four rows at one k
            const int mq = idx % (TM3 / 4), k = idx / (TM3 / 4);
            const int gm = min(m0 + 4 * mq, g.M - 4);
            pa[i] = A + (size_t)(kbeg + k) * g.sAk + gm;
            wa[i] = k * LDA + 4 * mq;
            ka[i] = k;
        }
    }
    const float* pb[NI];
    int wb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = t + 256 * i;
        if (BK) {  // B[n][k], k contiguous
            const int q = idx & 3, n = idx >> 2;
            const int gn = min(n0 + n, g.N - 1);
            pb[i] = Bm + (size_t)gn * g.sBn + 4 * q + kbeg;
            wb[i] = 4 * q * LDB + n;
            kb[i] = 4 * q;
        } else {   // B[k][n], n contiguous
            const int nq = idx % (TN3 / 4), k = idx / (TN3 / 4);
            const int gn = min(n0 + 4 * nq, g.N - 4);
            pb[i] = Bm + (size_t)(kbeg + k) * g.sBk + gn;
            wb[i] = k * LDB + 4 * nq;
            kb[i] = k;
        }
    }
    const size_t stepA = AK ? (size_t)TK : (size_t)TK * g.sAk;
    const size_t stepB = BK ? (size_t)TK : (size_t)TK * g.sBk;
    float4 ra[MI], rb[NI];
    // K tail (K % 16 != 0; k-contiguous operands: K % 4 == 0, so a float4 is inside or outside as a whole): an item past
    // K loads from the operand's base (valid, aligned) and is staged as ZERO
    bool kva[MI], kvb[NI];
    auto request = [&](int k0) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            kva[i] = k0 + ka[i] < kend;
            ra[i] = *(const float4*)(kva[i] ? pa[i] : A);
            pa[i] += stepA;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            kvb[i] = k0 + kb[i] < kend;
            rb[i] = *(const float4*)(kvb[i] ? pb[i] : Bm);
            pb[i] += stepB;
        }
    };
    // row sums of A over K (g.rowsum: the bias gradient of a weight-gradient launch, A = dz): a k-contiguous item is
    // four k of ONE row, the same row at every K step - the first column tile's workgroups add them up as they stage
    float rs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rs[i] = 0.f;
    const bool do_rs = AK && g.rowsum != nullptr && blockIdx.x == 0;
    auto stage = [&](int buf) {
        float* as = &As[buf][0][0];
        float* bs = &Bs[buf][0][0];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float4 u = kva[i] ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (AK) {
                if (do_rs) rs[i] += (u.x + u.y) + (u.z + u.w);
                as[wa[i]] = u.x; as[wa[i] + LDA] = u.y; as[wa[i] + 2 * LDA] = u.z; as[wa[i] + 3 * LDA] = u.w;
            } else {
                *(float4*)(as + wa[i]) = u;
            }
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 v = kvb[i] ? rb[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (BK) {
                bs[wb[i]] = v.x; bs[wb[i] + LDB] = v.y; bs[wb[i] + 2 * LDB] = v.z; bs[wb[i] + 3 * LDB] = v.w;
            } else {
                *(float4*)(bs + wb[i]) = v;
            }
        }
    };
    request(kbeg);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kbeg + ksteps * TK; k0 += TK) {
        const bool more = k0 + TK < kbeg + ksteps * TK;
        if (more) request(k0 + TK);
        const float* as = &As[buf][hi][32 * MI * wm + li];
        const float* bs = &Bs[buf][hi][32 * NI * wn + li];
#pragma unroll
        for (int kk = 0; kk < TK; kk += 2) {
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = as[kk * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = bs[kk * LDB + 32 * j];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);
        __syncthreads();  // tile t + 1 is staged, and every wave is done with tile t (the buffer tile t + 2 goes to)
        buf ^= 1;
    }
    if (KG > 1) {
        // the groups' partial tiles (and partial row sums) meet in the LDS: group kg > 0 parks its 16 accumulator values
        // per lane in ITS OWN stage memory (nobody else reads or writes it), group 0 adds them in group order
        float* park = &AsG[kg][0][0][0];  // 2 * TK * LDA = 2176 floats per As + the group's Bs behind it
        static_assert(2 * TK * LDA >= 8 * 256 + 128 && 2 * TK * LDB >= 8 * 256 + 128, "a group's stages hold its parked tile");
        float* parkB = &BsG[kg][0][0][0];
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* dst = r < 8 ? park + r * 256 : parkB + (r - 8) * 256;
                dst[t] = acc[0][0][r];
            }
            if (AK) (t < 128 ? park : parkB - 128)[8 * 256 + t] = rs[0];  // (2176 floats per array: 8 x 256 values + 128 spare)
        }
        __syncthreads();
        if (kg > 0) return;
        for (int o = 1; o < KG; ++o) {
            const float* pa_ = &AsG[o][0][0][0];
            const float* pb_ = &BsG[o][0][0][0];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += (r < 8 ? pa_ + r * 256 : pb_ + (r - 8) * 256)[t];
            if (AK) rs[0] += (t < 128 ? pa_ : pb_ - 128)[8 * 256 + t];
        }
    }
    if (AK && do_rs) {  // the four k-quads of a row sit in four neighbouring lanes
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            float v = rs[i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            const int m = (t + 256 * i) >> 2;
            if ((t & 3) == 0 && m0 + m < g.M) g.rowsum[(size_t)bz * g.bRowsum + m0 + m] = v;
        }
    }
    // epilogue: every load of an accumulator block (bias, the sigmoid factor's stored activation) is issued before the
    // first store, so 16 loads are in flight instead of one per round trip. Addresses = a wave-uniform row base (scalar
    // registers) + ONE 32-bit lane offset per matrix (generic3_ok: M x row stride < 2^31) - 64-bit addresses per element
    // cost an occupancy step.
    const float* bias = g.bias ? g.bias + (size_t)bz * g.bBias : nullptr;
    const float* Z = g.Z ? g.Z + (size_t)bz * g.bZ : nullptr;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int gn = n0 + 32 * NI * wn + 32 * j + li;
            const int rb0 = m0 + 32 * MI * wm + 32 * i + 4 * hi;  // this lane's first row of the block
            const unsigned zoff = (unsigned)rb0 * (unsigned)g.sZm + (unsigned)gn;
            const unsigned coff = (unsigned)rb0 * (unsigned)g.sCm + (unsigned)gn;
            const bool centre = !(g.eo_cols > 0 && gn >= g.eo_cols);
            float zv[16], bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                const bool ok = rb0 + ro < g.M && gn < g.N;
                bv[r] = (bias && ok) ? (bias + ro)[rb0] : 0.f;
                zv[r] = (g.sigmoid_mul && ok) ? (Z + (size_t)ro * g.sZm)[zoff] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] + (centre ? bv[r] : 0.f);
                if (g.sigmoid_mul) v *= nsvd_sigmoid_from_softplus(zv[r]);
                if (rb0 + ro < g.M && gn < g.N) (C + (size_t)ro * g.sCm)[coff] = v;
            }
            __builtin_amdgcn_sched_barrier(0);  // (the next block's loads stay behind this block's stores: registers)
        }
}

template <int MI, int NI, int KG = 1>
int launch_generic3(const NsvdGemm& g, hipStream_t s) {
    const bool ak = g.sAk == 1, bk = g.sBk == 1;
    dim3 grid(nsvd_cdiv(g.N, 64 * NI), nsvd_cdiv(g.M, 64 * MI), g.batch);
#define NSVD_G3(AKv, BKv) hipLaunchKernelGGL((gemm_generic3_kernel<MI, NI, AKv, BKv, KG>), grid, dim3(256 * KG), 0, s, g)
    if (ak && bk) NSVD_G3(true, true);
    else if (ak) NSVD_G3(true, false);
    else if (bk) NSVD_G3(false, true);
    else NSVD_G3(false, false);
#undef NSVD_G3
    NSVD_CHECK_LAUNCH();
    return 0;
}

// can the vectorised kernel take this launch? (fast axes contiguous and 16-byte aligned in every batch)
bool generic3_ok(const NsvdGemm& g) {
    auto al4 = [](long v) { return (v & 3) == 0; };
    auto alp = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (!alp(g.A) || !alp(g.B) || !al4(g.bA) || !al4(g.bB)) return false;
    if ((long)g.M * g.sCm >= (1L << 31) || (g.Z && (long)g.M * g.sZm >= (1L << 31))) return false;  // 32-bit lane offsets
    if (g.sAk == 1) { if (!al4(g.sAm) || !al4(g.K)) return false; }
    else if (g.sAm == 1) { if (!al4(g.sAk) || !al4(g.M) || g.M < 4) return false; }
    else return false;
    if (g.sBk == 1) { if (!al4(g.sBn) || !al4(g.K)) return false; }
    else if (g.sBn == 1) { if (!al4(g.sBk) || !al4(g.N) || g.N < 4) return false; }
    else return false;
    return true;
}

// In place: a layer's pre-activations -> activations (what the next layer's contraction, the weight gradient and the
// data gradient's sigmoid factor read). z: rows x R, R = nst * B columns in stencil blocks of B; block 0 = the centre
// evaluation -> softplus; blocks 1 + 2 d / 2 + 2 d = the even / odd perturbations along d -> the even /
// odd parts of softplus(z0 + zE +- zO) - softplus(z0) (nsvd_softplus_evenodd). A thread owns one (row, sample) - or
// four samples - in EVERY block: it reads the centre value before it overwrites it.
template <bool VEC>
__global__ void __launch_bounds__(256) softplus_inplace_kernel(float* __restrict__ z, long rows, int B, int nst, long R) {
    constexpr int W = VEC ? 4 : 1;
    const long per = B / W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * per) return;
    const long row = idx / per;
    const int b = (int)(idx - row * per) * W;
    float* p = z + row * R + b;
    float z0[W], a0[W];
    if constexpr (VEC) {
        const float4 v = *(const float4*)p;
        z0[0] = v.x; z0[1] = v.y; z0[2] = v.z; z0[W - 1] = v.w;
    } else {
        z0[0] = p[0];
    }
#pragma unroll
    for (int c = 0; c < W; ++c) a0[c] = nsvd_softplus(z0[c]);
    for (int d = 0; 2 * d + 2 < nst; ++d) {
        float* pe = p + (long)(1 + 2 * d) * B;
        float* po = p + (long)(2 + 2 * d) * B;
        float ze[W], zo[W];
        if constexpr (VEC) {
            const float4 e = *(const float4*)pe, o = *(const float4*)po;
            ze[0] = e.x; ze[1] = e.y; ze[2] = e.z; ze[W - 1] = e.w;
            zo[0] = o.x; zo[1] = o.y; zo[2] = o.z; zo[W - 1] = o.w;
        } else {
            ze[0] = pe[0]; zo[0] = po[0];
        }
#pragma unroll
        for (int c = 0; c < W; ++c) nsvd_softplus_evenodd(z0[c], ze[c], zo[c], &ze[c], &zo[c]);
        if constexpr (VEC) {
            *(float4*)pe = make_float4(ze[0], ze[1], ze[2], ze[W - 1]);
            *(float4*)po = make_float4(zo[0], zo[1], zo[2], zo[W - 1]);
        } else {
            pe[0] = ze[0]; po[0] = zo[0];
        }
    }
    if constexpr (VEC) *(float4*)p = make_float4(a0[0], a0[1], a0[2], a0[W - 1]);
    else p[0] = a0[0];
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

int nsvd_gemm_generic(const NsvdGemm& g, hipStream_t s, bool* rowsum_done) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || g.batch <= 0) return NSVD_EINVAL;
    if (rowsum_done) *rowsum_done = false;
    // the vectorised kernel where the launch allows it (NSVD_GEMM_GENERIC3=0: off, for A/B measurements)
    static const char* e3 = getenv("NSVD_GEMM_GENERIC3");
    if (!(e3 && e3[0] == '0') && generic3_ok(g)) {
        // the largest tile that still gives the chip enough workgroups (NSVD_G3_MINWG, default 1024: four per CU, what
        // the registers and the LDS let reside - measured against 128 / 256 / 512: hidden width 64 0.47 / 0.40 / 0.34 /
        // 0.34 ms per step, hidden width 128 0.65 / 0.62 / 0.53 / 0.47); a launch too small for any of them takes the
        // smallest tile
        if (rowsum_done) *rowsum_done = g.rowsum != nullptr && g.sAk == 1;  // (the k-contiguous-A instances add the row sums)
        const char* emw = getenv("NSVD_G3_MINWG");  // (read per launch: the tests switch tile shapes with it)
        const long minwg = emw ? atol(emw) : 1024;
        auto nwg = [&](int tm, int tn) { return (long)nsvd_cdiv(g.M, tm) * nsvd_cdiv(g.N, tn) * g.batch; };
        if (g.M > 64 && nwg(128, 128) >= minwg) return launch_generic3<2, 2>(g, s);
        if (g.N >= 256 && nwg(64, 256) >= minwg) return launch_generic3<1, 4>(g, s);
        if (nwg(64, 128) >= minwg) return launch_generic3<1, 2>(g, s);
        // fewer 64 x 64 tiles than half the CUs and a long contraction: four K groups per workgroup (NSVD_G3_KGROUPS=0: off)
        const char* ekg = getenv("NSVD_G3_KGROUPS");
        if (nwg(64, 64) <= 128 && g.K >= 256 && !(ekg && ekg[0] == '0')) return launch_generic3<1, 1, 4>(g, s);
        return launch_generic3<1, 1>(g, s);
    }
    dim3 grid2(nsvd_cdiv(g.N, T2N), nsvd_cdiv(g.M, T2M), g.batch);
    hipLaunchKernelGGL(gemm_generic2_kernel, grid2, dim3(256), 0, s, g);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_softplus_inplace(float* z, long rows, int B, int nst, hipStream_t s) {
    if (!z || rows <= 0 || B <= 0 || nst < 1 || (nst > 1 && (nst & 1) == 0)) return NSVD_EINVAL;
    const long R = (long)nst * B;
    const bool vec = (B & 3) == 0 && ((uintptr_t)z & 15) == 0;
    const long n = rows * (vec ? B / 4 : B);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (vec) hipLaunchKernelGGL(softplus_inplace_kernel<true>, dim3(blocks), dim3(256), 0, s, z, rows, B, nst, R);
    else hipLaunchKernelGGL(softplus_inplace_kernel<false>, dim3(blocks), dim3(256), 0, s, z, rows, B, nst, R);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_rowsum(const float* in, float* out, int rows, int n, long ld, hipStream_t s) {
    const int blocks = nsvd_cdiv(rows, 4);
    hipLaunchKernelGGL(rowsum_kernel, dim3(blocks), dim3(256), 0, s, in, out, rows, n, ld);
    NSVD_CHECK_LAUNCH();
    return 0;
}
