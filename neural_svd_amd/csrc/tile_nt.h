// C = A B^T tile routine on the fp32-input MFMA, both operands K-contiguous: one 64 x 64 tile per 256-thread block.
// Shared by the CDK loss contractions (cdk_loss.hip) and the hidden-layer weight gradients (pmlp_bwd.hip).
#pragma once
#include "nsvd_common.h"

typedef float nsvd_f32x16 __attribute__((ext_vector_type(16)));

constexpr int NSVD_TNT_T = 64;                 // tile edge (rows of A x rows of B per block of 4 waves)
constexpr int NSVD_TNT_KC = 64;                // contraction chunk
constexpr int NSVD_TNT_LDT = NSVD_TNT_KC + 4;  // padded LDS row: conflict-free ds_read_b128 over 32 rows
constexpr int NSVD_TNT_BUF = 2 * NSVD_TNT_T * NSVD_TNT_LDT;  // one buffer: A chunk then B chunk
constexpr int NSVD_TNT_FLOATS = 2 * NSVD_TNT_BUF;            // double-buffered: 68 KB

__device__ __forceinline__ float nsvd_tnt_sum4(float4 v) { return (v.x + v.y) + (v.z + v.w); }

// acc(32x32 of this wave) += A[64 rows][k0 : k1] . B[64 rows][k0 : k1]^T for the 64 x 64 tile of a 256-thread
// block (k1 - k0 a multiple of 64). Wave wv owns rows 32 (wv & 1) of A and rows 32 (wv >> 1) of B.
// Global loads run TWO chunks ahead through two named register sets (the operands were written by other XCDs:
// MALL / HBM latency); LDS is double-buffered with one barrier per chunk. The loop body is branch-free (loads
// past the end re-read the last chunk; their LDS copy is never consumed) so that the compiler's waitcnt counting
// keeps the younger set in flight. Four independent accumulation chains (k mod 4), summed at the end.
// Measured with s_memtime at K = 576 (9 chunks, one block per CU): 28.8 K cycles per tile, of which 20.8 K are
// the 288 MFMAs themselves; the rest is the per-chunk write / barrier / first-read latency that a single wave
// per SIMD cannot hide.
// ROWSUM: also return, in rsum[i], this thread's share of the row sums of A over [k0, k1) (tile rows
// (t >> 4) + 16 i, columns 4 (t & 15) .. +3 of every chunk): reduce over the 16 threads that share a row.
#define NSVD_TNT_LOAD(S, ko)                                         \
    sa##S##0 = *(const float4*)(ap0 + (ko));                         \
    sa##S##1 = *(const float4*)(ap1 + (ko));                         \
    sa##S##2 = *(const float4*)(ap2 + (ko));                         \
    sa##S##3 = *(const float4*)(ap3 + (ko));                         \
    sb##S##0 = *(const float4*)(bp + (ko));                          \
    sb##S##1 = *(const float4*)(bp + 16 * ldb + (ko));               \
    sb##S##2 = *(const float4*)(bp + 32 * ldb + (ko));               \
    sb##S##3 = *(const float4*)(bp + 48 * ldb + (ko));
#define NSVD_TNT_PUT(S, buf)                                                  \
    {                                                                         \
        float* la_ = lds + (buf) * NSVD_TNT_BUF + lr0 * NSVD_TNT_LDT + lc;    \
        float* lb_ = la_ + NSVD_TNT_T * NSVD_TNT_LDT;                         \
        *(float4*)(la_) = sa##S##0;                                           \
        *(float4*)(la_ + 16 * NSVD_TNT_LDT) = sa##S##1;                       \
        *(float4*)(la_ + 32 * NSVD_TNT_LDT) = sa##S##2;                       \
        *(float4*)(la_ + 48 * NSVD_TNT_LDT) = sa##S##3;                       \
        *(float4*)(lb_) = sb##S##0;                                           \
        *(float4*)(lb_ + 16 * NSVD_TNT_LDT) = sb##S##1;                       \
        *(float4*)(lb_ + 32 * NSVD_TNT_LDT) = sb##S##2;                       \
        *(float4*)(lb_ + 48 * NSVD_TNT_LDT) = sb##S##3;                       \
    }
#define NSVD_TNT_SUM(S)                                                       \
    {                                                                         \
        rsum[0] += nsvd_tnt_sum4(sa##S##0);                                   \
        rsum[1] += nsvd_tnt_sum4(sa##S##1);                                   \
        rsum[2] += nsvd_tnt_sum4(sa##S##2);                                   \
        rsum[3] += nsvd_tnt_sum4(sa##S##3);                                   \
    }
// 32 MFMAs on LDS buffer `buf`; the reads of group s+1 are issued ahead of the 4 MFMAs of group s
#define NSVD_TNT_COMPUTE(buf)                                                                 \
    {                                                                                         \
        const float* la = lds + (buf) * NSVD_TNT_BUF + ra * NSVD_TNT_LDT + kq;                \
        const float* lb = lds + (buf) * NSVD_TNT_BUF + NSVD_TNT_T * NSVD_TNT_LDT + rb * NSVD_TNT_LDT + kq; \
        float4 av = *(const float4*)la, bv = *(const float4*)lb;                              \
        _Pragma("unroll") for (int s = 0; s < NSVD_TNT_KC / 8; ++s) {                         \
            float4 an = av, bn = bv;                                                          \
            if (s + 1 < NSVD_TNT_KC / 8) {                                                    \
                an = *(const float4*)(la + (s + 1) * 8);                                      \
                bn = *(const float4*)(lb + (s + 1) * 8);                                      \
            }                                                                                 \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);             \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc1, 0, 0, 0);           \
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc2, 0, 0, 0);           \
            acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc3, 0, 0, 0);           \
            av = an;                                                                          \
            bv = bn;                                                                          \
        }                                                                                     \
    }
// workgroup barrier that leaves the global loads in flight (__syncthreads() waits for vmcnt(0): the chunk requested at
// the top of the step would have to land by its end - one step of lead instead of the two the register sets are for)
#define NSVD_TNT_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// one pipeline step: prefetch chunk c+2 into set S, compute chunk c, stage chunk c+1 (set SN) into LDS
#define NSVD_TNT_STEP(S, SN, c)                                                               \
    NSVD_TNT_LOAD(S, min((c) + 2, nc - 1) * NSVD_TNT_KC)                                      \
    NSVD_TNT_COMPUTE(S)                                                                       \
    __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);                                        \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                        \
    _Pragma("unroll") for (int s = 0; s + 1 < NSVD_TNT_KC / 8; ++s) {                         \
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                    \
    }                                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                        \
    NSVD_TNT_PUT(SN, SN)                                                                      \
    if (ROWSUM && (c) + 1 < nc) NSVD_TNT_SUM(SN)                                              \
    NSVD_TNT_BARRIER();

// core: the four A rows this thread stages (tile rows (t >> 4) + 16 i) are given as pointers, already offset to
// its 4 columns of the first chunk - the rows of A may therefore be GATHERED (kernel_apply.hip); bp likewise.
template <bool ROWSUM>
__device__ __forceinline__ void nsvd_tile_nt_rows(const float* __restrict__ ap0, const float* __restrict__ ap1,
                                                  const float* __restrict__ ap2, const float* __restrict__ ap3,
                                                  const float* __restrict__ bp, long ldb, int nc,
                                                  float* __restrict__ lds, nsvd_f32x16& acc, float (&rsum)[4]) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int ra = (wv & 1) * 32 + (lane & 31), rb = (wv >> 1) * 32 + (lane & 31), kq = (lane >> 5) * 4;
    const int lr0 = t >> 4, lc = (t & 15) * 4;
    // named registers + macros: hipcc demotes a conditionally rewritten float4 array to scratch
    float4 sa00, sa01, sa02, sa03, sb00, sb01, sb02, sb03;  // set 0: even chunks -> LDS buffer 0
    float4 sa10, sa11, sa12, sa13, sb10, sb11, sb12, sb13;  // set 1: odd chunks  -> LDS buffer 1
    nsvd_f32x16 acc1 = {0}, acc2 = {0}, acc3 = {0};
    NSVD_TNT_LOAD(0, 0)
    NSVD_TNT_LOAD(1, min(1, nc - 1) * NSVD_TNT_KC)
    NSVD_TNT_PUT(0, 0)
    if (ROWSUM) NSVD_TNT_SUM(0)
    __syncthreads();
    for (int c = 0; c < nc; c += 2) {
        NSVD_TNT_STEP(0, 1, c)
        if (c + 1 >= nc) break;
        NSVD_TNT_STEP(1, 0, c + 1)
    }
    acc = (acc + acc1) + (acc2 + acc3);
}

template <bool ROWSUM>
__device__ __forceinline__ void nsvd_tile_nt(const float* __restrict__ A, long lda, const float* __restrict__ Bm,
                                             long ldb, int k0, int k1, float* __restrict__ lds, nsvd_f32x16& acc,
                                             float (&rsum)[4]) {
    const int t = threadIdx.x;
    const int lr0 = t >> 4, lc = (t & 15) * 4;  // staging: rows lr0 + 16 i, 16 floats per row-quarter
    const float* ap = A + (long)lr0 * lda + lc + k0;
    nsvd_tile_nt_rows<ROWSUM>(ap, ap + 16 * lda, ap + 32 * lda, ap + 48 * lda, Bm + (long)lr0 * ldb + lc + k0, ldb,
                              (k1 - k0) / NSVD_TNT_KC, lds, acc, rsum);
}
#undef NSVD_TNT_LOAD
#undef NSVD_TNT_PUT
#undef NSVD_TNT_SUM
#undef NSVD_TNT_COMPUTE
#undef NSVD_TNT_STEP
#undef NSVD_TNT_BARRIER
