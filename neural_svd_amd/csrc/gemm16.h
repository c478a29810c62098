// C = A B^T-style contractions on the bf16 MFMA for the mixed-precision CDK towers (tower16.hip) and
// nsvd_gemm_bf16 (include/nsvd.h): 256 x 128 output tile per workgroup, K in steps of 64, 8 waves (2 per SIMD, 4 x 2,
// 64 x 64 per wave = 4 x 4 blocks of v_mfma_f32_16x16x32_bf16), operands global -> LDS by LDS-DMA
// (global_load_lds_dwordx4) into a ring of three stages, one workgroup barrier per K step, the DMA of step t + 2 in
// flight under the MFMAs of steps t and t + 1. Up to four independent problems per launch (the two towers; both weight-
// gradient contractions of both towers: one kernel boundary instead of two), each with its own output shape.
// Reference arithmetic this replaces: the five matmuls per tower and step of examples/models/mlp.py:129-164 under
// torch.cuda.amp.autocast (examples/cdk/sketchy/main_sketchy.py:182) and their autograd backward.
//
// Operand forms. Each operand is either k-contiguous ("T": row r of the tile is a row of the matrix, K values
// contiguous - X, W in the forward) or k-strided ("S": the matrix is stored (K, rows), rows contiguous - the batch-
// contracted weight gradients dW = dY^T A read dY and A as they are, and dA = dY W reads W as it is): no transposed
// copies exist anywhere in the mixed-precision tower.
//   T image  [rows][64 k] bf16, 128-byte rows; the 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7): the
//            ds_read_b128 fragment reads of 16 consecutive rows x one chunk pair hit 16 distinct slots of the 256-byte
//            bank row (MI355X_MICROARCH.md, LDS: lane groups of ds_read_b128).
//   S image  [64 k][128 rows] bf16 per 128-row block, 256-byte rows; chunk c of k-row q sits in slot
//            c ^ (((q & 3) << 2) | ((q >> 2) & 3)); fragments by ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane
//            group, delivered column-major: cdna_hip_programming.md T10, image (b)) - two reads per fragment.
//   The swizzles are applied on the DMA's per-lane SOURCE address (the LDS destination of an LDS-DMA is lane-linear)
//   and again on the fragment reads.
// MFMA orientation: D = mfma(B fragment, A fragment) - the accumulator's lane index is the tile ROW (of A), its four
// registers four consecutive COLUMNS: one 16-byte (float32) / 8-byte (bfloat16) store per block and lane.
#pragma once
#include <stdlib.h>
#include "nsvd_common.h"

namespace nsvd_g16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int BM = 256, BN = 128, BK = 64, NST = 3;
constexpr int A_BYTES = BM * BK * 2;         // 32 KB
constexpr int B_BYTES = BN * BK * 2;         // 16 KB
constexpr int ST_BYTES = A_BYTES + B_BYTES;  // 48 KB
constexpr int LDS_BYTES = NST * ST_BYTES;    // 144 KB
constexpr int NDMA = (A_BYTES + B_BYTES) / 1024 / 8;  // LDS-DMA instructions per wave and stage: 6

struct Prob {
    const void* A;      // bf16: T form (M, lda), S form (K, lda)
    const void* B;      // bf16: T form (N, ldb), S form (K, ldb)
    void* C;            // (M, ldc) float32 or bfloat16; split-K slice s at C + s * slice_stride elements
    const float* bias;  // per column of C (N) or null
    float* sumsq;       // null, or one float per workgroup of THIS problem: the sum of squares of its tile of C
    int M, N;           // this problem's output shape (the problems of a launch share K, S and the operand forms)
    long lda, ldb, ldc; // in elements
    int tiles_m, tiles_n, wg0;  // filled by launch(): tile counts; this problem's first workgroup in the launch's order
};

constexpr int MAXPROB = 4;

struct Args {
    Prob p[MAXPROB];
    int nprob, K, S;            // S split-K slices of K / S each
    long slice_stride;          // elements of C between split-K slices
    int nwg;
    int f16;  // the operands (and a 16-bit output) are IEEE float16 instead of bfloat16
    int dbg;  // diagnostic builds only (NSVD_G16_DBG): 1 = no DMA after the prologue, 2 = no fragment reads / MFMAs
    unsigned long long* stamps;  // diagnostic, or null: cycles of block 0 / wave 0 - prologue, K loop (first half | barrier | second half), epilogue
};

// the problems of a launch with one shape (the two towers): shape and strides of all nprob entries
inline void set_uniform(Args& a, int M, int N, long lda, long ldb, long ldc) {
    for (int i = 0; i < a.nprob; ++i) {
        a.p[i].M = M; a.p[i].N = N; a.p[i].lda = lda; a.p[i].ldb = ldb; a.p[i].ldc = ldc;
    }
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
// The HALF TYPE of a mixed-precision launch: bfloat16 (F16 = false), or IEEE float16 (F16 = true: the reference's
// autocast dtype, examples/cdk/sketchy/main_sketchy.py:182 - used with the loss scaling of cdk_step.hip). Two 16-bit
// values of either type from two floats (round to nearest even), and back; the MFMA on a pair of 8-value fragments.
template <bool F16>
__device__ __forceinline__ unsigned pack_h(float a, float b) {
    if (F16) return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, f16x2));
    return pack_bf16(a, b);
}
template <bool F16>
__device__ __forceinline__ float h_lo(unsigned u) {
    if (F16) return (float)__builtin_bit_cast(f16x2, u)[0];
    return __uint_as_float(u << 16);
}
template <bool F16>
__device__ __forceinline__ float h_hi(unsigned u) {
    if (F16) return (float)__builtin_bit_cast(f16x2, u)[1];
    return __uint_as_float(u & 0xffff0000u);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8& b, const bf16x8& a, const f32x4& c) {
    if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, b), __builtin_bit_cast(f16x8, a), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
}

// 16 B per lane global -> LDS: source = sbase (wave-uniform) + voff (per lane, bytes), destination = m0 (wave-uniform LDS
// byte address) + lane * 16. M0 is written here behind the compiler's back (check_m0.py: nothing else in the
// translation unit may use it).
__device__ __forceinline__ void dma16(const char* sbase, unsigned voff, unsigned m0) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0), "v"(voff), "s"(sbase) : "memory");
}

template <bool AS, bool BS, bool O16, bool F16 = false>
__global__ void __launch_bounds__(512) gemm16_kernel(Args a) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    // ---- which tile: XCD-aware order (workgroups b, b + 8, .. share an XCD and its L2: they get a contiguous run of the
    // launch's order - problem, split-K slice, then the tiles with the fast index over the dimension with fewer tiles)
    int id = blockIdx.x;
    if ((a.nwg & 7) == 0) id = (id & 7) * (a.nwg >> 3) + (id >> 3);
    int prob = 0;
#pragma unroll
    for (int i = 1; i < MAXPROB; ++i)
        if (i < a.nprob && id >= a.p[i].wg0) prob = i;
    const Prob& P = a.p[prob];
    id -= P.wg0;
    const int per = P.tiles_m * P.tiles_n;
    const int slice = id / per;
    id -= slice * per;
    int tm, tn;
    if (P.tiles_m <= P.tiles_n) {
        tn = id / P.tiles_m;
        tm = id - tn * P.tiles_m;
    } else {
        tm = id / P.tiles_n;
        tn = id - tm * P.tiles_n;
    }
    const int Ks = a.K / a.S;
    const int nk = Ks / BK;
    const long k0 = (long)slice * Ks;
    const unsigned lda = (unsigned)P.lda, ldb = (unsigned)P.ldb;

    // ---- DMA sources: wave w moves pieces 4 w .. 4 w + 3 of A and 2 w, 2 w + 1 of B (1 KB each)
    const char* sa = reinterpret_cast<const char*>(P.A) + 2 * (AS ? (k0 * P.lda + (long)BM * tm) : ((long)BM * tm * P.lda + k0));
    const char* sb = reinterpret_cast<const char*>(P.B) + 2 * (BS ? (k0 * P.ldb + (long)BN * tn) : ((long)BN * tn * P.ldb + k0));
    const long sa_step = AS ? 2L * BK * P.lda : 2L * BK;
    const long sb_step = BS ? 2L * BK * P.ldb : 2L * BK;
    unsigned va[4], vb[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = 4 * w + j;
        if (AS) {
            const int krow = 4 * (p & 15) + (lane >> 4), slot = lane & 15;
            const int chunk = slot ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
            va[j] = 2u * ((unsigned)krow * lda + 128u * (unsigned)(p >> 4) + 8u * (unsigned)chunk);
        } else {
            const int row = 8 * p + (lane >> 3), slot = lane & 7;
            const int chunk = slot ^ ((row >> 1) & 7);
            va[j] = 2u * ((unsigned)row * lda + 8u * (unsigned)chunk);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = 2 * w + j;
        if (BS) {
            const int krow = 4 * p + (lane >> 4), slot = lane & 15;
            const int chunk = slot ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
            vb[j] = 2u * ((unsigned)krow * ldb + 8u * (unsigned)chunk);
        } else {
            const int row = 8 * p + (lane >> 3), slot = lane & 7;
            const int chunk = slot ^ ((row >> 1) & 7);
            vb[j] = 2u * ((unsigned)row * ldb + 8u * (unsigned)chunk);
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned m0a = lds0 + 4096u * (unsigned)w, m0b = lds0 + A_BYTES + 2048u * (unsigned)w;
#define G16_ISSUE(stage)                                               \
    {                                                                  \
        const unsigned so = (unsigned)(stage) * (unsigned)ST_BYTES;    \
        dma16(sa, va[0], m0a + so);                                    \
        dma16(sa, va[1], m0a + so + 1024u);                            \
        dma16(sa, va[2], m0a + so + 2048u);                            \
        dma16(sa, va[3], m0a + so + 3072u);                            \
        dma16(sb, vb[0], m0b + so);                                    \
        dma16(sb, vb[1], m0b + so + 1024u);                            \
        sa += sa_step;                                                 \
        sb += sb_step;                                                 \
    }
#define G16_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")

    // ---- fragment addresses (byte offsets inside a stage)
    // T image: block i (16 rows), k32 step s: row r0 + 16 i + (lane & 15), chunk 4 s + (lane >> 4)
    // S image: block i, k32 step s, half h2: k-row 32 s + 8 (lane >> 4) + 4 h2 + q, q = (lane & 15) >> 2;
    //          columns mo + 4 (lane & 3) .. + 3 of the 128-row block
    const int l15 = lane & 15, g4 = lane >> 4;
    int fa[2] = {0, 0}, fb[2] = {0, 0};  // T images: [k32 step s], blocks at + 2048 i (the S addresses: in the loop)
    if (!AS) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            fa[s] = (64 * wm + l15) * 128 + (((4 * s + g4) ^ ((l15 >> 1) & 7)) << 4);
    }
    if (!BS) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
            fb[s] = A_BYTES + (64 * wn + l15) * 128 + (((4 * s + g4) ^ ((l15 >> 1) & 7)) << 4);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragment loads. T image: block i at + 2048 i from fa[s] / fb[s]. S image: per block and k-row half one
    // address (the block enters through its chunk index INSIDE the swizzle), the k32 step at + 8192.
#define G16_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr))
    int sa_off[4][2], sb_off[4][2];  // S images: [block][k-row half]
    if (AS || BS) {
        const int q = l15 >> 2, pp = l15 & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int krow = 8 * g4 + 4 * h2 + q;
                const int sw = ((krow & 3) << 2) | ((krow >> 2) & 3);
                const int ca = ((64 * (wm & 1) + 16 * i) >> 3) + (pp >> 1);
                const int cb = ((64 * wn + 16 * i) >> 3) + (pp >> 1);
                sa_off[i][h2] = 16384 * (wm >> 1) + 256 * krow + 16 * (ca ^ sw) + 8 * (pp & 1);
                sb_off[i][h2] = A_BYTES + 256 * krow + 16 * (cb ^ sw) + 8 * (pp & 1);
            }
    }
    struct Frags {
        bf16x8 a[4], b[4];
    };
    auto load_frags = [&](Frags& f, const char* st, int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AS) {
                const s16x4 lo = G16_TR(st + sa_off[i][0] + 8192 * s), hi = G16_TR(st + sa_off[i][1] + 8192 * s);
                f.a[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                f.a[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + fa[s] + 2048 * i));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BS) {
                const s16x4 lo = G16_TR(st + sb_off[j][0] + 8192 * s), hi = G16_TR(st + sb_off[j][1] + 8192 * s);
                f.b[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                f.b[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + fb[s] + 2048 * j));
            }
        }
    };
    auto mma = [&](const Frags& f) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = mfma16<F16>(f.b[j], f.a[i], acc[i][j]);
    };

    // ---- the K loop, software-pipelined across the K steps: ONE workgroup barrier per step, in its MIDDLE.
    //   first half of step t : MFMAs on the fragments of (t, k32 step 0) | reads of (t, k32 step 1)
    //   barrier t            : "stage t + 1 has landed" (counted vmcnt: the DMA of t + 2 stays in flight) - and every
    //                          wave has read all of stage t (its reads waited for: lgkmcnt(0))
    //   DMA of step t + 3 into stage t % 3 (just freed)
    //   second half          : MFMAs on (t, 1) | reads of (t + 1, 0)
    // Every MFMA block runs on fragments read half a step earlier; a DMA has two steps to land.
    const bool stamp = a.stamps && blockIdx.x == 0 && tid == 0;
    const unsigned long long ts0 = a.stamps ? __builtin_readcyclecounter() : 0ull;
    unsigned long long th1 = 0, thb = 0, th2 = 0;
    G16_ISSUE(0);
    if (nk > 1) G16_ISSUE(1);
    if (nk > 2) G16_ISSUE(2);
    if (nk > 2) G16_WAIT_BARRIER(2 * NDMA);
    else if (nk > 1) G16_WAIT_BARRIER(NDMA);
    else G16_WAIT_BARRIER(0);
    // the second-dispatched half of the workgroup loses every issue arbitration to its SIMD partner (same program, older
    // wave first): a static priority for waves 4-7 evens that out (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (w >= 4 && !(a.dbg & 16)) __builtin_amdgcn_s_setprio(1);
    Frags f0, f1;
    if (!(a.dbg & 2)) load_frags(f0, lds, 0);
    const unsigned long long ts1 = a.stamps ? __builtin_readcyclecounter() : 0ull;
    for (int t = 0; t < nk; ++t) {
        const char* st = lds + (t % NST) * ST_BYTES;
        const char* sn = lds + ((t + 1) % NST) * ST_BYTES;
        const unsigned long long u0 = a.stamps ? __builtin_readcyclecounter() : 0ull;
        if (!(a.dbg & 2)) {
            load_frags(f1, st, 1);
            mma(f0);
        }
        const unsigned long long u1 = a.stamps ? __builtin_readcyclecounter() : 0ull;
        if (t + 2 < nk && !(a.dbg & 1)) G16_WAIT_BARRIER(NDMA);
        else G16_WAIT_BARRIER(0);
        const unsigned long long u2 = a.stamps ? __builtin_readcyclecounter() : 0ull;
        if (t + 3 < nk && !(a.dbg & 1)) G16_ISSUE(t % NST);
        if (!(a.dbg & 2)) {
            if (t + 1 < nk) load_frags(f0, sn, 0);
            mma(f1);
        }
        const unsigned long long u3 = a.stamps ? __builtin_readcyclecounter() : 0ull;
        th1 += u1 - u0; thb += u2 - u1; th2 += u3 - u2;
    }
    const unsigned long long ts2 = a.stamps ? __builtin_readcyclecounter() : 0ull;
#undef G16_ISSUE
#undef G16_WAIT_BARRIER
#undef G16_TR

    // ---- epilogue: lane = tile row 64 wm + 16 i + (lane & 15); registers = columns 64 wn + 16 j + 4 (lane >> 4) ..+3.
    // Straight from the registers a store instruction would write 16 rows x 64 (32) bytes - a quarter of a cache line
    // per row, 5.5 us for the 16 MB of a (1024, 8192) bfloat16 output. The tile goes through LDS instead (the ring is
    // free now) and leaves as whole rows: 512 (256) contiguous bytes per row, 16-byte stores.
    constexpr int RS = O16 ? 272 : 528;  // staged row stride in bytes (padded: the column writes spread over the banks)
    float ss = 0.f;
    __syncthreads();  // every wave is done reading the last stage
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.bias) bv = *reinterpret_cast<const float4*>(P.bias + BN * tn + 64 * wn + 4 * g4 + 16 * j);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float c0 = acc[i][j][0] + bv.x, c1 = acc[i][j][1] + bv.y, c2 = acc[i][j][2] + bv.z,
                        c3 = acc[i][j][3] + bv.w;
            ss = fmaf(c0, c0, ss); ss = fmaf(c1, c1, ss); ss = fmaf(c2, c2, ss); ss = fmaf(c3, c3, ss);
            char* dst = lds + (64 * wm + 16 * i + l15) * RS + (64 * wn + 16 * j + 4 * g4) * (O16 ? 2 : 4);
            if (a.dbg & 4) {
                asm volatile("" ::"v"(c0), "v"(c1), "v"(c2), "v"(c3));
            } else if (O16) {
                *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h<F16>(c0, c1), pack_h<F16>(c2, c3));
            } else {
                *reinterpret_cast<float4*>(dst) = make_float4(c0, c1, c2, c3);
            }
        }
    }
    __syncthreads();
    if (!(a.dbg & 4)) {
        constexpr int LPR = O16 ? 16 : 32;        // lanes per row (16 bytes each)
        constexpr int RPP = 512 / LPR;            // rows per pass of the workgroup
        const int rr = tid / LPR, cc = tid % LPR;
        char* cbase = reinterpret_cast<char*>(P.C) + ((long)slice * a.slice_stride + (long)BM * tm * P.ldc + BN * tn) * (O16 ? 2 : 4);
        for (int r = rr; r < BM; r += RPP) {
            const uint4 v = *reinterpret_cast<const uint4*>(lds + r * RS + 16 * cc);
            char* dstp = cbase + (long)r * P.ldc * (O16 ? 2 : 4) + 16 * cc;
            // write-through (sc1): the rows leave for memory as they are stored instead of sitting dirty in this XCD's L2
            // until the kernel-end write-back (MI355X_MICROARCH.md, stores of each flavour / row `boundary`): 0.9 us of
            // the 15.8 at (1024, 8192, 512) bfloat16 out, 1.5 of 23.5 at (512, 8192, 1024) float32 out
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            // (s_nop 1 inside the string: hipcc pads nothing behind an asm store - its next instruction may overwrite the
            // data registers before the store has read them: intermittent wrong elements, found in round 6)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstp), "v"(vv) : "memory");
        }
    }
    if (stamp) {
        a.stamps[0] = ts1 - ts0; a.stamps[1] = th1 / nk; a.stamps[2] = thb / nk; a.stamps[3] = th2 / nk;
        a.stamps[4] = __builtin_readcyclecounter() - ts2; a.stamps[5] = (unsigned long long)nk;
    }
    if (P.sumsq) {  // fixed order: lanes of a wave (butterfly), then the eight waves
        ss = nsvd_wave_sum(ss);
        __syncthreads();  // the staged tile has been read out
        float* red = reinterpret_cast<float*>(lds);
        if (lane == 0) red[w] = ss;
        __syncthreads();
        if (tid == 0)
            P.sumsq[slice * per + tm * P.tiles_n + tn] =
                ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
}

// the two-workgroups-per-CU form of the kernel (gemm16b.h, included at the end of this header)
template <bool AS_, bool BS_, bool O_, bool F_>
inline int launch_b_inst(const Args& a, hipStream_t s);

// host side: validate and launch. out_bf16: C holds bfloat16. Returns 0 or NSVD_E*.
inline int launch(const Args& a0, bool a_strided, bool b_strided, bool out_bf16, hipStream_t s) {
    Args a = a0;
    if (a.nprob < 1 || a.nprob > MAXPROB || a.S < 1 || a.K <= 0 || a.K % (BK * a.S)) return NSVD_EINVAL;
    int nwg = 0;
    for (int i = 0; i < a.nprob; ++i) {
        Prob& P = a.p[i];
        if (P.M <= 0 || P.N <= 0 || P.M % BM || P.N % BN) return NSVD_EINVAL;
        if ((P.lda % 8) || (P.ldb % 8) || (P.ldc % (out_bf16 ? 8 : 4))) return NSVD_EINVAL;  // 16-byte DMA sources and stores
        if (!P.A || !P.B || !P.C) return NSVD_EINVAL;
        if (((uintptr_t)P.A | (uintptr_t)P.B | (uintptr_t)P.C | (uintptr_t)P.bias) & 15) return NSVD_EINVAL;
        // per-lane source offsets are 32-bit: a tile's rows must lie within 4 GB of its origin
        const long span_a = 2L * (a_strided ? (long)BK * P.lda + BM : (long)BM * P.lda + BK);
        const long span_b = 2L * (b_strided ? (long)BK * P.ldb + BN : (long)BN * P.ldb + BK);
        if (span_a >= (1L << 32) || span_b >= (1L << 32)) return NSVD_EINVAL;
        P.tiles_m = P.M / BM;
        P.tiles_n = P.N / BN;
        P.wg0 = nwg;
        nwg += P.tiles_m * P.tiles_n * a.S;
    }
    a.nwg = nwg;
    {
        static const char* e = getenv("NSVD_G16_DBG");
        a.dbg = e ? atoi(e) : 0;
    }
    const dim3 grid((unsigned)nwg);
    static const char* form_env = getenv("NSVD_G16_FORM");  // "a": the one-workgroup-per-CU kernel above (A/B measurements)
    // two workgroups per CU (gemm16b.h) where the launch has at least two per CU to give; one per CU otherwise (this kernel:
    // deeper K steps, fragments double-buffered - better alone on its CU)
    const bool form_b = !(form_env && form_env[0] == 'a') && !a.stamps && nwg >= 512;
#define G16_LAUNCH_T(AS_, BS_, O_, F_)                                                                             \
    if (form_b) {                                                                                                  \
        const int rcb = launch_b_inst<AS_, BS_, O_, F_>(a, s);                                                     \
        if (rcb) return rcb;                                                                                       \
    } else {                                                                                                       \
        static bool attr_set = false;                                                                              \
        if (!attr_set) {                                                                                           \
            hipError_t e = hipFuncSetAttribute((const void*)gemm16_kernel<AS_, BS_, O_, F_>,                       \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);             \
            if (e != hipSuccess) return -(int)e;                                                                   \
            attr_set = true;                                                                                       \
        }                                                                                                          \
        hipLaunchKernelGGL((gemm16_kernel<AS_, BS_, O_, F_>), grid, dim3(512), LDS_BYTES, s, a);                   \
    }
#define G16_LAUNCH(AS_, BS_, O_)                                                                                   \
    {                                                                                                              \
        if (a.f16) { G16_LAUNCH_T(AS_, BS_, O_, true) } else { G16_LAUNCH_T(AS_, BS_, O_, false) }                 \
    }
    if (!a_strided && !b_strided) {
        if (out_bf16) G16_LAUNCH(false, false, true) else G16_LAUNCH(false, false, false)
    } else if (!a_strided && b_strided) {
        if (out_bf16) G16_LAUNCH(false, true, true) else G16_LAUNCH(false, true, false)
    } else if (a_strided && b_strided) {
        if (out_bf16) G16_LAUNCH(true, true, true) else G16_LAUNCH(true, true, false)
    } else {
        return NSVD_EUNSUPPORTED;  // (A k-strided, B k-contiguous): no contraction of the towers has this form
    }
#undef G16_LAUNCH_T
#undef G16_LAUNCH
    NSVD_CHECK_LAUNCH();
    return 0;
}

}  // namespace nsvd_g16

#include "gemm16b.h"
