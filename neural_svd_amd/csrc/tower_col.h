// The wide layer of a mixed-precision CDK tower with BatchNorm INSIDE the contraction's epilogue (round 6): a workgroup
// owns 64 whole COLUMNS of the (B, d1) layer - all B <= 1024 rows, the accumulators of the whole column block in the
// registers of its 8 waves (128 rows x 64 columns = 128 registers per lane at B = 1024) - so the per-column batch
// statistics are a reduction inside the workgroup and the pre-normalisation output never exists in memory:
//   forward   A1h  = lrelu(BN1(Xh W1h^T + b1))                one launch; stores A1h (bfloat16), mean, 1 / std
//   backward  dY1h = BN1'(lrelu'(dY2h W2h))                    one launch; reads A1h, stores dY1h, dgamma, dbeta, db1
// replacing gemm16 + tower_bn16_forward and gemm16 + tower_bn16_backward of tower.hip (and the (B, d1) round trips of
// Y1h and dA1h between them: 4 x 16.8 MB written and read per step at configs[4]).
// Reference arithmetic: Linear -> BatchNorm1d (training) -> LeakyReLU of examples/models/mlp.py:129-164 under autocast
// (examples/cdk/sketchy/main_sketchy.py:182) and its autograd backward.
//
// What is rounded where (the float64 oracle restates exactly this: oracle.tower_forward_backward(gemm_bf16="fused")):
// Xh, W1h, W2h, dY2h bfloat16 operands as before; Y1 = Xh W1h^T + b1 stays float32 in the accumulators (statistics,
// normalisation and activation on the unrounded values); A1 is stored as bfloat16. The backward has no Y1 to read: it
// recovers the normalised value from the STORED activation, h = A1 > 0 ? A1 : A1 / slope, yhat = (h - beta) / gamma
// (leaky ReLU is invertible for slope > 0: the "in-place activated BatchNorm" identity) - so this form needs slope > 0
// and gamma != 0 (tower.hip keeps the strip kernels for slope == 0); dA1 = dY2h W2h stays float32 in the accumulators;
// dY1 is stored as bfloat16, the bias / BatchNorm-weight gradients are float32 column sums of the unrounded values.
//
// Loop: K in stages of 64 (one 128-byte line of bfloat16 per row), both operands global -> LDS by LDS-DMA in full lines
// (cdna_hip_programming.md, projection GEMM with whole columns per block: "x through LDS in full lines"). A wave's 16 NI
// rows are ITS OWN (wave w: rows w RW ..): the A image is wave-private, [rows][64 k] with gemm16.h's T swizzle, staged
// PI row blocks (16 PI rows) per pass - NP = NI / PI passes over K with the accumulators of all passes kept; the
// 64-column weight tile (8 KB per stage) is shared: one workgroup barrier per stage. B operand: forward W1 (N, K)
// k-contiguous (T image, ds_read_b128); backward W2 (K, N) as stored, k-strided: an S image [64 k][64 n] of 128-byte
// rows read by ds_read_b64_tr_b16, chunk c of k-row q in slot c ^ 2 (((q >> 1) & 1) | (((q >> 3) & 1) << 1)): the 32 lanes
// of a half (k-rows q .. q + 3 and q + 8 .. q + 11, four 8-byte pieces each) then hit 32 distinct 8-byte slots.
#pragma once
#include "gemm16.h"

namespace nsvd_tcol {

using nsvd_g16::bf16x8;
using nsvd_g16::dma16;
using nsvd_g16::f32x4;
using nsvd_g16::h_hi;
using nsvd_g16::h_lo;
using nsvd_g16::mfma16;
using nsvd_g16::pack_h;
using nsvd_g16::s16x4;
typedef unsigned short bf16_t;

constexpr int TN = 64;                   // columns per workgroup
constexpr int BK = 64;                   // k per stage
constexpr int B_BYTES = TN * BK * 2;     // 8 KB
constexpr int RED_BYTES = 8 * 128 * 4 + 3 * 64 * 4;  // cross-wave partials [8][128] + three 64-column rows

struct Prob {
    const bf16_t* A;       // (M, K) k-contiguous: Xh (forward), dY2h (backward)
    const bf16_t* W;       // forward: W1h (N, K); backward: W2h (K, N)
    const float* bias;     // forward: b1 (N)
    const float* gamma;    // (N)
    const float* beta;     // (N)
    float* running_mean;   // forward: updated in place, or null
    float* running_var;
    float* mean;           // forward: written (N)
    float* invstd;         // forward: written; backward: read
    bf16_t* out;           // forward: A1h (M, N); backward: dY1h (M, N)
    const bf16_t* A1;      // backward: the stored activation (M, N)
    float* dgamma;         // backward (N)
    float* dbeta;
    float* dbias;
    float* sumsq;          // backward: null, or one float per workgroup of this tower: sum of squares of the 3 x 64 small
                           // gradients it wrote
};

struct Args {
    Prob p[2];
    int nt, M, N, K;
    float eps, momentum, slope;
    int nwg;
    int f16;  // the half type is IEEE float16 instead of bfloat16 (gemm16.h: pack_h)
    unsigned long long* stamps;  // diagnostic, or null: cycles of block 0 / wave 0 - prologue, K loop, epilogue
};

// sum over the 16 lanes that share lane >> 4 (the 16 rows of a block), fixed butterfly order; every lane gets the sum
__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 8, 64);
    return v;
}

template <bool BWD, int NI, int PI, int NBUF, bool F16 = false>
__global__ void __launch_bounds__(512) tower_col_kernel(Args a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    constexpr int NP = NI / PI;
    constexpr int RW = 16 * NI;             // rows per wave
    constexpr int A_WAVE = 16 * PI * 128;   // bytes of a wave's A image per stage
    constexpr int A_BYTES = 8 * A_WAVE;
    constexpr int ST = A_BYTES + B_BYTES;
    constexpr int NDMA = 2 * PI + 1;        // LDS-DMA instructions per wave and stage
    static_assert(NI % PI == 0 && NBUF >= 2 && NBUF <= 3, "tower_col: shape");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g4 = lane >> 4;

    // XCD-aware order: workgroups b, b + 8, .. share an XCD; they get a contiguous run of (tower, column tile), so an
    // XCD's L2 holds ONE tower's A operand (1 MB at configs[4]) for its 32 tiles
    int id = blockIdx.x;
    if ((a.nwg & 7) == 0) id = (id & 7) * (a.nwg >> 3) + (id >> 3);
    const int tiles = a.N / TN;
    const int tw = id / tiles;
    const int tn = id - tw * tiles;
    const Prob& P = a.p[tw];
    const int n0 = tn * TN;
    const int K = a.K, N = a.N, M = a.M;
    const int nkt = K / BK;
    const int NS = NP * nkt;

    const unsigned long long ts0 = a.stamps ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long wc0 = a.stamps ? wall_clock64() : 0ull;

    // ---- DMA sources
    const char* sa = reinterpret_cast<const char*>(P.A) + 2L * (long)(w * RW) * K;
    const long sa_pass = 2L * K * (16 * PI) - 2L * K;  // after the last stage of a pass: next 16 PI rows, k = 0
    unsigned va[2 * PI];
#pragma unroll
    for (int j = 0; j < 2 * PI; ++j) {
        const int rowl = 8 * j + (lane >> 3), slot = lane & 7;
        const int chunk = slot ^ ((rowl >> 1) & 7);
        va[j] = 2u * ((unsigned)rowl * (unsigned)K + 8u * (unsigned)chunk);
    }
    const char* sb0;
    long sb_step;
    unsigned vb;
    {
        const int r8 = 8 * w + (lane >> 3), slot = lane & 7;
        if (BWD) {  // k-rows of W2 (K, N): 64 columns = one 128-byte line per k-row
            const int chunk = slot ^ ((((r8 >> 1) & 1) | (((r8 >> 3) & 1) << 1)) << 1);
            sb0 = reinterpret_cast<const char*>(P.W) + 2L * n0;
            sb_step = 2L * BK * N;
            vb = 2u * ((unsigned)r8 * (unsigned)N + 8u * (unsigned)chunk);
        } else {    // rows n of W1 (N, K)
            const int chunk = slot ^ ((r8 >> 1) & 7);
            sb0 = reinterpret_cast<const char*>(P.W) + 2L * (long)n0 * K;
            sb_step = 2L * BK;
            vb = 2u * ((unsigned)r8 * (unsigned)K + 8u * (unsigned)chunk);
        }
    }
    const char* sb = sb0;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned m0a = lds0 + (unsigned)A_WAVE * (unsigned)w;
    const unsigned m0b = lds0 + A_BYTES + 1024u * (unsigned)w;
    int iss_t = 0, iss_buf = 0;
#define TC_ISSUE()                                                                  \
    {                                                                               \
        const unsigned so = (unsigned)iss_buf * (unsigned)ST;                       \
        _Pragma("unroll") for (int j = 0; j < 2 * PI; ++j) dma16(sa, va[j], m0a + so + 1024u * (unsigned)j); \
        dma16(sb, vb, m0b + so);                                                    \
        sa += 2 * BK;                                                               \
        sb += sb_step;                                                              \
        iss_buf = iss_buf + 1 == NBUF ? 0 : iss_buf + 1;                            \
        if (++iss_t == nkt) {                                                       \
            iss_t = 0;                                                              \
            sa += sa_pass;                                                          \
            sb = sb0;                                                               \
        }                                                                           \
    }
#define TC_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
#define TC_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr))

    // ---- fragment offsets inside a stage
    int fa[2], fb[2];  // T images, k32 step s; row blocks at + 2048 i
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        fa[s] = A_WAVE * w + l15 * 128 + (((4 * s + g4) ^ ((l15 >> 1) & 7)) << 4);
        fb[s] = A_BYTES + l15 * 128 + (((4 * s + g4) ^ ((l15 >> 1) & 7)) << 4);
    }
    // S image (backward): k-row 32 s + 8 g4 + 4 h2 + q, q = l15 >> 2; 8-byte piece pp = l15 & 3 of the block's 32 bytes
    const int q = l15 >> 2, pp = l15 & 3;
    const int fk = (((q >> 1) & 1) | ((g4 & 1) << 1)) << 1;
    int sbase[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) sbase[h2] = A_BYTES + (8 * g4 + 4 * h2 + q) * 128 + 8 * (pp & 1);

    f32x4 acc[NI][4];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: NBUF - 1 stages in flight
    TC_ISSUE();
    if (NBUF == 3 && NS > 1) TC_ISSUE();
    const unsigned long long ts1 = a.stamps ? __builtin_readcyclecounter() : 0ull;

    int step = 0, buf = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        for (int t = 0; t < nkt; ++t, ++step) {
            // stage `step` has landed (younger stages may still fly) and every wave is done with stage step - 1
            if (NBUF == 3 && step + 1 < NS) TC_WAIT_BARRIER(NDMA);
            else TC_WAIT_BARRIER(0);
            if (step + NBUF - 1 < NS) TC_ISSUE();
            const char* st = lds + buf * ST;
            buf = buf + 1 == NBUF ? 0 : buf + 1;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 fa_[PI], fb_[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BWD) {
                        const int co = 16 * (((2 * j) ^ fk) + (pp >> 1)) + 4096 * s;
                        const s16x4 lo = TC_TR(st + sbase[0] + co), hi = TC_TR(st + sbase[1] + co);
                        fb_[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                    } else {
                        fb_[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + fb[s] + 2048 * j));
                    }
                }
#pragma unroll
                for (int i = 0; i < PI; ++i)
                    fa_[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + fa[s] + 2048 * i));
#pragma unroll
                for (int i = 0; i < PI; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[p * PI + i][j] = mfma16<F16>(fb_[j], fa_[i], acc[p * PI + i][j]);
            }
        }
    }
#undef TC_ISSUE
#undef TC_TR
    const unsigned long long ts2 = a.stamps ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long wc1 = a.stamps ? wall_clock64() : 0ull;

    // ---- epilogue. lane = row 16 i + l15 of the wave's rows; registers = columns 16 j + 4 g4 .. + 3 of the tile.
    __syncthreads();  // every wave is done reading the last stage: the ring is free
    char* img = lds + (size_t)w * (RW * 128);                         // this wave's [RW][64] bfloat16 image, T swizzle
    float* red = reinterpret_cast<float*>(lds + 8 * RW * 128);        // [8 waves][128]
    float* crow = red + 8 * 128;                                      // [3][64] column rows
    const float rM = 1.0f / (float)M;
    // image address of this lane's 8 bytes (columns 16 j + 4 g4 .. + 3) of row 16 i + l15: + 2048 i
    int ia[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ia[j] = l15 * 128 + (((2 * j + (g4 >> 1)) ^ ((l15 >> 1) & 7)) << 4) + 8 * (g4 & 1);

    // column totals of per-lane partials v[j][e] (NV = 1) or of two sets at once (NV = 2: second set in v2):
    // 16 lanes of a row block by butterfly, the 8 waves through LDS in wave order by one thread per column; the totals
    // come back in crow[0] (and crow[1]); two barriers, the results valid until the next call's first barrier
    auto col_totals = [&](float (&v)[4][4], float (*v2)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] = row16_sum(v[j][e]);
                if (v2) v2[j][e] = row16_sum(v2[j][e]);
            }
        if (l15 == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<float4*>(red + w * 128 + 16 * j + 4 * g4) = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                if (v2)
                    *reinterpret_cast<float4*>(red + w * 128 + 64 + 16 * j + 4 * g4) =
                        make_float4(v2[j][0], v2[j][1], v2[j][2], v2[j][3]);
            }
        }
        __syncthreads();
        if (tid < (v2 ? 128 : 64)) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
            crow[tid] = t;  // (tid 64 .. 127: crow[1])
        }
        __syncthreads();
    };
    auto read_row = [&](const float* r, float (&o)[4][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(r + 16 * j + 4 * g4);
            o[j][0] = t.x; o[j][1] = t.y; o[j][2] = t.z; o[j][3] = t.w;
        }
    };
    // the wave's image out as whole rows: 8 rows x 128 bytes per store instruction, write-through
    auto store_image = [&]() {
        bf16_t* out = P.out;
#pragma unroll 4
        for (int it = 0; it < RW / 8; ++it) {
            const int r = 8 * it + (lane >> 3), slot = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(img + r * 128 + 16 * slot);
            const int chunk = slot ^ ((r >> 1) & 7);
            char* dst = reinterpret_cast<char*>(out) + ((long)(w * RW + r) * N + n0) * 2 + 16 * chunk;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            // (s_nop 1 inside the string: hipcc pads nothing behind an asm store - its next instruction may overwrite the
            // data registers before the store has read them: intermittent wrong elements, found in round 6)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(vv) : "memory");
        }
    };

    if (!BWD) {
        // y = acc + b1; two-pass batch statistics on the unrounded values
        float s1[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 bv = *reinterpret_cast<const float4*>(P.bias + n0 + 16 * j + 4 * g4);
            s1[j][0] = s1[j][1] = s1[j][2] = s1[j][3] = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                acc[i][j][0] += bv.x; acc[i][j][1] += bv.y; acc[i][j][2] += bv.z; acc[i][j][3] += bv.w;
#pragma unroll
                for (int e = 0; e < 4; ++e) s1[j][e] += acc[i][j][e];
            }
        }
        col_totals(s1, nullptr);
        if (tid < 64) crow[128 + tid] = crow[tid] * rM;  // crow[2]: the mean (kept for the statistics rows)
        float mu[4][4];
        read_row(crow, mu);
        float s2[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mu[j][e] *= rM;
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const float d = acc[i][j][e] - mu[j][e];
                    t = fmaf(d, d, t);
                }
                s2[j][e] = t;
            }
        col_totals(s2, nullptr);
        if (tid < 64) {
            const float var = crow[tid], cm = crow[128 + tid];
            const float inv = 1.0f / sqrtf(var * rM + a.eps);
            P.mean[n0 + tid] = cm;
            P.invstd[n0 + tid] = inv;
            if (P.running_mean) {
                const float unb = var / (float)(M - 1);
                P.running_mean[n0 + tid] = (1.f - a.momentum) * P.running_mean[n0 + tid] + a.momentum * cm;
                P.running_var[n0 + tid] = (1.f - a.momentum) * P.running_var[n0 + tid] + a.momentum * unb;
            }
        }
        float var[4][4];
        read_row(crow, var);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(P.gamma + n0 + 16 * j + 4 * g4);
            const float4 ev = *reinterpret_cast<const float4*>(P.beta + n0 + 16 * j + 4 * g4);
            const float ga[4] = {gv.x, gv.y, gv.z, gv.w}, be[4] = {ev.x, ev.y, ev.z, ev.w};
            float inv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) inv[e] = 1.0f / sqrtf(var[j][e] * rM + a.eps);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = fmaf((acc[i][j][e] - mu[j][e]) * inv[e], ga[e], be[e]);
                    o[e] = t > 0.f ? t : a.slope * t;
                }
                *reinterpret_cast<uint2*>(img + 2048 * i + ia[j]) = make_uint2(pack_h<F16>(o[0], o[1]), pack_h<F16>(o[2], o[3]));
            }
        }
        __syncthreads();
        store_image();
    } else {
        // the stored activation's tile into this wave's image (full lines, swizzled on the source address)
        {
            const char* src = reinterpret_cast<const char*>(P.A1) + ((long)(w * RW) * N + n0) * 2;
            const unsigned m0i = lds0 + (unsigned)w * (unsigned)(RW * 128);
#pragma unroll 4
            for (int it = 0; it < RW / 8; ++it) {
                const int r = 8 * it + (lane >> 3), slot = lane & 7;
                const int chunk = slot ^ ((r >> 1) & 7);
                dma16(src, 2u * ((unsigned)r * (unsigned)N + 8u * (unsigned)chunk), m0i + 1024u * (unsigned)it);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const float rslope = 1.0f / a.slope;
        // per-lane partial sums of 4 columns -> this wave's row of `red` (set k: columns 64 k + ..): the 16 lanes of a
        // row block by butterfly; few registers live at a time (the accumulators fill half the file)
        auto put_partial = [&](int j, int set, const float (&v)[4]) {
            const float t0 = row16_sum(v[0]), t1 = row16_sum(v[1]), t2 = row16_sum(v[2]), t3 = row16_sum(v[3]);
            if (l15 == 0) *reinterpret_cast<float4*>(red + w * 128 + 64 * set + 16 * j + 4 * g4) = make_float4(t0, t1, t2, t3);
        };
        // the 8 waves' rows added in wave order by one thread per column -> crow[set]; two barriers
        auto finish_totals = [&](int nsets) {
            __syncthreads();
            if (tid < 64 * nsets) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += red[k * 128 + tid];
                crow[tid] = t;
            }
            __syncthreads();
        };
        // pass 1: dh = dA1 * lrelu'(h) (kept in the accumulators); column sums of dh and dh * yhat
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(P.gamma + n0 + 16 * j + 4 * g4);
            const float4 ev = *reinterpret_cast<const float4*>(P.beta + n0 + 16 * j + 4 * g4);
            const float rg[4] = {1.0f / gv.x, 1.0f / gv.y, 1.0f / gv.z, 1.0f / gv.w}, be[4] = {ev.x, ev.y, ev.z, ev.w};
            float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const uint2 u = *reinterpret_cast<const uint2*>(img + 2048 * i + ia[j]);
                const float av[4] = {h_lo<F16>(u.x), h_hi<F16>(u.x), h_lo<F16>(u.y), h_hi<F16>(u.y)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool pos = av[e] > 0.f;
                    const float h = pos ? av[e] : av[e] * rslope;
                    const float yh = (h - be[e]) * rg[e];
                    const float dh = acc[i][j][e] * (pos ? 1.f : a.slope);
                    acc[i][j][e] = dh;
                    s1[e] += dh;
                    s2[e] = fmaf(dh, yh, s2[e]);
                }
            }
            put_partial(j, 0, s1);
            put_partial(j, 1, s2);
        }
        finish_totals(2);
        float qsum = 0.f;  // (sum of the squares of dbeta, dgamma: taken now, crow is overwritten by the next reduction)
        if (tid < 64) {
            const float c1 = crow[tid], c2 = crow[64 + tid];
            P.dbeta[n0 + tid] = c1;
            P.dgamma[n0 + tid] = c2;
            qsum = fmaf(c1, c1, c2 * c2);
        }
        // pass 2: dy = gamma inv (dh - mean(dh) - yhat mean(dh yhat)); stored over the activation's image
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 gv = *reinterpret_cast<const float4*>(P.gamma + n0 + 16 * j + 4 * g4);
            const float4 ev = *reinterpret_cast<const float4*>(P.beta + n0 + 16 * j + 4 * g4);
            const float4 iv = *reinterpret_cast<const float4*>(P.invstd + n0 + 16 * j + 4 * g4);
            const float4 c1 = *reinterpret_cast<const float4*>(crow + 16 * j + 4 * g4);
            const float4 c2 = *reinterpret_cast<const float4*>(crow + 64 + 16 * j + 4 * g4);
            const float be[4] = {ev.x, ev.y, ev.z, ev.w};
            const float gi[4] = {gv.x * iv.x, gv.y * iv.y, gv.z * iv.z, gv.w * iv.w};
            const float rg[4] = {1.0f / gv.x, 1.0f / gv.y, 1.0f / gv.z, 1.0f / gv.w};
            const float m1[4] = {c1.x * rM, c1.y * rM, c1.z * rM, c1.w * rM};
            const float m2[4] = {c2.x * rM, c2.y * rM, c2.z * rM, c2.w * rM};
            float s3[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                char* cell = img + 2048 * i + ia[j];
                const uint2 u = *reinterpret_cast<const uint2*>(cell);
                const float av[4] = {h_lo<F16>(u.x), h_hi<F16>(u.x), h_lo<F16>(u.y), h_hi<F16>(u.y)};
                float dy[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float h = av[e] > 0.f ? av[e] : av[e] * rslope;
                    const float yh = (h - be[e]) * rg[e];
                    dy[e] = gi[e] * (acc[i][j][e] - m1[e] - yh * m2[e]);
                    s3[e] += dy[e];
                }
                *reinterpret_cast<uint2*>(cell) = make_uint2(pack_h<F16>(dy[0], dy[1]), pack_h<F16>(dy[2], dy[3]));
            }
            put_partial(j, 0, s3);
        }
        // (the readers of crow above are past their reads before anyone overwrites it: the first barrier below)
        finish_totals(1);  // (its barriers also order the image writes before store_image's reads)
        if (tid < 64) {
            const float c3 = crow[tid];
            P.dbias[n0 + tid] = c3;
            if (P.sumsq) {  // tid < 64 = wave 0: butterfly = fixed order
                qsum = fmaf(c3, c3, qsum);
                qsum = nsvd_wave_sum(qsum);
                if (tid == 0) P.sumsq[tn] = qsum;
            }
        }
        store_image();
    }
    if (a.stamps && blockIdx.x == 0 && tid == 0) {
        a.stamps[0] = ts1 - ts0; a.stamps[1] = ts2 - ts1; a.stamps[2] = __builtin_readcyclecounter() - ts2;
        a.stamps[3] = (unsigned long long)NS;
    }
    if (a.stamps) {  // every block: start / loop end / end on the 100 MHz clock (waves may end apart: the last one wins)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            if (w == 0) { a.stamps[4 + 4 * blockIdx.x] = wc0; a.stamps[5 + 4 * blockIdx.x] = wc1; }
            atomicMax(&a.stamps[6 + 4 * blockIdx.x], wall_clock64());
        }
    }
#undef TC_WAIT_BARRIER
}

// can this launch take the shape? M rows in {256, 512, 768, 1024}, N % 64 == 0, K % 64 == 0; per-lane 32-bit source offsets
inline bool shape_ok(int M, int N, int K) {
    return M >= 256 && M <= 1024 && M % 256 == 0 && N > 0 && N % TN == 0 && K > 0 && K % BK == 0 &&
           128L * K * 2 < (1L << 31) && 64L * (long)N * 2 < (1L << 31) && 128L * (long)N * 2 < (1L << 31);
}

template <bool BWD, int NI, int PI, int NBUF, bool F16>
inline int launch_inst_t(const Args& a, hipStream_t s) {
    constexpr int ring = NBUF * (8 * 16 * PI * 128 + B_BYTES);
    constexpr int epi = 8 * 16 * NI * 128 + RED_BYTES;
    constexpr int lds_bytes = ring > epi ? ring : epi;
    static_assert(lds_bytes <= 160 * 1024, "tower_col: LDS");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)tower_col_kernel<BWD, NI, PI, NBUF, F16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((tower_col_kernel<BWD, NI, PI, NBUF, F16>), dim3((unsigned)a.nwg), dim3(512), lds_bytes, s, a);
    return 0;
}

template <bool BWD, int NI, int PI, int NBUF>
inline int launch_inst(const Args& a, hipStream_t s) {
    return a.f16 ? launch_inst_t<BWD, NI, PI, NBUF, true>(a, s) : launch_inst_t<BWD, NI, PI, NBUF, false>(a, s);
}

template <bool BWD>
inline int launch(const Args& a0, hipStream_t s) {
    Args a = a0;
    if (a.nt < 1 || a.nt > 2 || !shape_ok(a.M, a.N, a.K)) return NSVD_EINVAL;
    for (int t = 0; t < a.nt; ++t) {
        const Prob& P = a.p[t];
        if (!P.A || !P.W || !P.gamma || !P.beta || !P.invstd || !P.out) return NSVD_EINVAL;
        if (BWD ? (!P.A1 || !P.dgamma || !P.dbeta || !P.dbias) : (!P.bias || !P.mean)) return NSVD_EINVAL;
        if (((uintptr_t)P.A | (uintptr_t)P.W | (uintptr_t)P.out | (uintptr_t)P.A1 | (uintptr_t)P.bias | (uintptr_t)P.gamma |
             (uintptr_t)P.beta | (uintptr_t)P.invstd) & 15)
            return NSVD_EINVAL;
    }
    a.nwg = a.nt * (a.N / TN);
    // stages: at 128 rows per wave two passes of 4 row blocks with two 72 KB buffers, or four passes of 2 with three
    // 40 KB buffers (NSVD_TCOL_FORM=b)
    static const char* form = getenv("NSVD_TCOL_FORM");
    const bool fb = form && form[0] == 'b';
    int rc;
    switch (a.M / 128) {
        case 2: rc = launch_inst<BWD, 2, 2, 3>(a, s); break;
        case 4: rc = fb ? launch_inst<BWD, 4, 2, 3>(a, s) : launch_inst<BWD, 4, 4, 2>(a, s); break;
        case 6: rc = launch_inst<BWD, 6, 3, 2>(a, s); break;
        default: rc = fb ? launch_inst<BWD, 8, 2, 3>(a, s) : launch_inst<BWD, 8, 4, 2>(a, s); break;
    }
    if (rc) return rc;
    NSVD_CHECK_LAUNCH();
    return 0;
}

}  // namespace nsvd_tcol
