// C (128 x 128) += A (128 rows) . B (128 rows)^T over a run of 32-wide contraction chunks, both operands contraction-
// contiguous ("NT"), fp32-input MFMA: the K loop shared by the layer-0 weight gradient (pmlp_bwd.hip) and the tower
// GEMMs (tower.hip). 256 threads = 4 waves as 2 x 2 of 64 x 64 (wave (wm, wn) owns rows 64 wm of A x rows 64 wn of B:
// acc[i][j] = its 32 x 32 block (32 i, 32 j)).
//   a_src / b_src: this thread's first float4 of chunk 0: row (tid >> 3) of the tile, columns 4 (tid & 7)..;
//   a_step / b_step: floats between tile rows r and r + 32; chunk c starts at column 32 c.
//   As / Bs: two stage buffers of 128 rows x A_LD floats each (2 x 128 x 36 x 4 B = 36 KB per operand).
//   rs: row sums of the A rows this thread stages (rows (tid >> 3) + 32 j, its 4 columns of every chunk).
// Software pipeline: fragments one q-group ahead, chunk c+1 written to the other LDS buffer under chunk c's third
// q-group, chunk c+2 fetched from global under its fourth (after the barrier), every memory instruction in an MFMA gap
// (sched_group_barrier), branch-free steady state. Measured: 72 K cycles for 16 chunks (65.5 K of MFMA issue).
#pragma once
#include "pmlp_common.h"

namespace nsvd_pmlp {

__device__ __forceinline__ void nsvd_tile128_nt(const float* a_src, const float* b_src, size_t a_step, size_t b_step,
                                                int nch, float* As, float* Bs, f32x16 (&acc)[2][2], float (&rs)[4]) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int s_row = tid >> 3, s_c4 = tid & 7;
#define WG_LD(dst, src) dst = *reinterpret_cast<const float4*>(src)
#define WG_ST(dst, v) *reinterpret_cast<float4*>(dst) = (v)
    // Same software pipeline as the forward's layer 0: fragments one q-group ahead, chunk c+1 written to
    // the other LDS buffer under chunk c's third q-group, chunk c+2 fetched from global under its fourth
    // (after the barrier), every memory instruction in an MFMA gap, branch-free steady state.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;  // row sums of the A rows this thread stages
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define WA_LOAD(c)                                                 \
    {                                                              \
        const float* pa_ = a_src + (c) * BK;                       \
        const float* pb_ = b_src + (c) * BK;                       \
        WG_LD(ra0, pa_);                                           \
        WG_LD(ra1, pa_ + a_step);                                    \
        WG_LD(ra2, pa_ + 2 * a_step);                                \
        WG_LD(ra3, pa_ + 3 * a_step);                                \
        WG_LD(rb0, pb_);                                           \
        WG_LD(rb1, pb_ + b_step);                                    \
        WG_LD(rb2, pb_ + 2 * b_step);                                \
        WG_LD(rb3, pb_ + 3 * b_step);                                \
    }
#define WA_STORE(buf)                                                              \
    {                                                                              \
        float* Ab_ = As + (buf) * HID * A_LD + s_row * A_LD + 4 * s_c4;            \
        float* Bb_ = Bs + (buf) * HID * A_LD + s_row * A_LD + 4 * s_c4;            \
        WG_ST(Ab_, ra0);                                                           \
        WG_ST(Ab_ + 32 * A_LD, ra1);                                               \
        WG_ST(Ab_ + 64 * A_LD, ra2);                                               \
        WG_ST(Ab_ + 96 * A_LD, ra3);                                               \
        WG_ST(Bb_, rb0);                                                           \
        WG_ST(Bb_ + 32 * A_LD, rb1);                                               \
        WG_ST(Bb_ + 64 * A_LD, rb2);                                               \
        WG_ST(Bb_ + 96 * A_LD, rb3);                                               \
        rs0 += (ra0.x + ra0.y) + (ra0.z + ra0.w);                                  \
        rs1 += (ra1.x + ra1.y) + (ra1.z + ra1.w);                                  \
        rs2 += (ra2.x + ra2.y) + (ra2.z + ra2.w);                                  \
        rs3 += (ra3.x + ra3.y) + (ra3.z + ra3.w);                                  \
    }
    struct F4 {
        float4 a0, a1, b0, b1;
    };
#define WA_READ(f, Ap, Bp, q)                                                      \
    {                                                                              \
        f.a0 = *reinterpret_cast<const float4*>((Ap) + 8 * (q));                   \
        f.a1 = *reinterpret_cast<const float4*>((Ap) + 32 * A_LD + 8 * (q));       \
        f.b0 = *reinterpret_cast<const float4*>((Bp) + 8 * (q));                   \
        f.b1 = *reinterpret_cast<const float4*>((Bp) + 32 * A_LD + 8 * (q));       \
    }
#define WA_MMA1(f, X)                                                                                   \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b0.X, acc[0][0], 0, 0, 0);              \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b1.X, acc[0][1], 0, 0, 0);              \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b0.X, acc[1][0], 0, 0, 0);              \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b1.X, acc[1][1], 0, 0, 0);
#define WA_MMA(f) WA_MMA1(f, x) WA_MMA1(f, y) WA_MMA1(f, z) WA_MMA1(f, w)
#define WA_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WA_IL(n, mask)                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier((mask), 1, 0);        \
    }
#define WA_BODY(c, DO_STORE, DO_LOAD)                                                           \
    {                                                                                           \
        const int cur = (c) & 1;                                                                \
        const float* Ap = As + cur * HID * A_LD + (64 * wm + li) * A_LD + 4 * hi;               \
        const float* Bp = Bs + cur * HID * A_LD + (64 * wn + li) * A_LD + 4 * hi;               \
        WA_READ(f1, Ap, Bp, 1);                                                                 \
        WA_MMA(f0);                                                                             \
        WA_IL(4, 0x100);                                                                        \
        WA_FENCE();                                                                             \
        WA_READ(f0, Ap, Bp, 2);                                                                 \
        WA_MMA(f1);                                                                             \
        WA_IL(4, 0x100);                                                                        \
        WA_FENCE();                                                                             \
        WA_READ(f1, Ap, Bp, 3);                                                                 \
        if (DO_STORE) WA_STORE(cur ^ 1);                                                        \
        WA_MMA(f0);                                                                             \
        WA_IL(4, 0x100);                                                                        \
        if (DO_STORE) WA_IL(8, 0x200);                                                          \
        WA_FENCE();                                                                             \
        __syncthreads();                                                                        \
        if (DO_STORE) {                                                                         \
            const float* An = As + (cur ^ 1) * HID * A_LD + (64 * wm + li) * A_LD + 4 * hi;     \
            const float* Bn = Bs + (cur ^ 1) * HID * A_LD + (64 * wn + li) * A_LD + 4 * hi;     \
            WA_READ(f0, An, Bn, 0);                                                             \
        }                                                                                       \
        if (DO_LOAD) WA_LOAD((c) + 2);                                                          \
        WA_MMA(f1);                                                                             \
        if (DO_STORE) WA_IL(4, 0x100);                                                          \
        if (DO_LOAD) WA_IL(8, 0x020);                                                           \
        WA_FENCE();                                                                             \
    }
    WA_LOAD(0);
    WA_STORE(0);
    __syncthreads();
    if (nch > 1) WA_LOAD(1);
    F4 f0, f1;
    {
        const float* Ap = As + (64 * wm + li) * A_LD + 4 * hi;
        const float* Bp = Bs + (64 * wn + li) * A_LD + 4 * hi;
        WA_READ(f0, Ap, Bp, 0);
    }
    {
        int c = 0;
        for (; c + 2 < nch; ++c) WA_BODY(c, true, true)
        if (c + 1 < nch) {
            WA_BODY(c, true, false)
            ++c;
        }
        WA_BODY(c, false, false)
    }
#undef WA_BODY
#undef WA_IL
#undef WA_FENCE
#undef WA_MMA
#undef WA_MMA1
#undef WA_READ
#undef WA_LOAD
#undef WA_STORE
#undef WG_LD
#undef WG_ST
    rs[0] = rs0; rs[1] = rs1; rs[2] = rs2; rs[3] = rs3;
}

}  // namespace nsvd_pmlp
