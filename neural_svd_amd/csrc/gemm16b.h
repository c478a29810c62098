// The bf16 contraction of gemm16.h with TWO workgroups per CU: the same 256 x 128 tile and operand forms, K in steps of 32
// (one v_mfma_f32_16x16x32_bf16 k-block per stage), a ring of three 24 KB stages (72 KB), fragments single-buffered and
// the kernel held to 128 registers, so that two 512-thread workgroups share a CU (four waves per SIMD). Why: stamped,
// a tile of gemm16.h spends 2.9 K cycles in its prologue, 1.67 K per K step of 1.02 K MFMA issue and 6.2 K in its
// epilogue - an HBM-write burst of the whole chip - with nothing else on the CU to cover any of it (gemm16.h);
// here the co-resident workgroup's K loop runs under the other's prologue, barrier waits and output burst.
//   T image  [rows][32 k] bf16, 64-byte rows; chunk c (16 B) of row r sits in slot c ^ (((r >> 3) & 1) << 1): the
//            ds_read_b128 lane groups {rows 0-3, 12-15 with chunk g | rows 4-11 with chunk g ^ 1} then hit 16 distinct
//            16-byte slots of the 256-byte bank row.
//   S image  [32 k][128 rows] bf16 per 128-row block (gemm16.h's S image, half the k-rows), read by ds_read_b64_tr_b16.
// Same arithmetic as gemm16.h (accumulation order over k included): same bits.
#pragma once
#include "gemm16.h"

namespace nsvd_g16 {

constexpr int B_BK = 32, B_NST = 3;
constexpr int B_A_BYTES = BM * B_BK * 2;             // 16 KB
constexpr int B_B_BYTES = BN * B_BK * 2;             // 8 KB
constexpr int B_ST_BYTES = B_A_BYTES + B_B_BYTES;    // 24 KB
constexpr int B_LDS_BYTES = B_NST * B_ST_BYTES;      // 72 KB
constexpr int B_NDMA = 3;                            // LDS-DMA instructions per wave and stage (2 of A, 1 of B)

template <bool AS, bool BS, bool O16, bool F16 = false>
__global__ void __launch_bounds__(512, 4) gemm16b_kernel(Args a) {  // (4 waves per SIMD: 128 registers, two workgroups per CU)
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
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
    const int nk = Ks / B_BK;
    const long k0 = (long)slice * Ks;
    const unsigned lda = (unsigned)P.lda, ldb = (unsigned)P.ldb;

    // ---- DMA sources: wave w moves pieces 2 w, 2 w + 1 of A and piece w of B (1 KB each)
    const char* sa = reinterpret_cast<const char*>(P.A) + 2 * (AS ? (k0 * P.lda + (long)BM * tm) : ((long)BM * tm * P.lda + k0));
    const char* sb = reinterpret_cast<const char*>(P.B) + 2 * (BS ? (k0 * P.ldb + (long)BN * tn) : ((long)BN * tn * P.ldb + k0));
    const long sa_step = AS ? 2L * B_BK * P.lda : 2L * B_BK;
    const long sb_step = BS ? 2L * B_BK * P.ldb : 2L * B_BK;
    unsigned va[2], vb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = 2 * w + j;
        if (AS) {
            const int krow = 4 * (p & 7) + (lane >> 4), slot = lane & 15;
            const int chunk = slot ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
            va[j] = 2u * ((unsigned)krow * lda + 128u * (unsigned)(p >> 3) + 8u * (unsigned)chunk);
        } else {
            const int row = 16 * p + (lane >> 2), slot = lane & 3;
            const int chunk = slot ^ (((row >> 3) & 1) << 1);
            va[j] = 2u * ((unsigned)row * lda + 8u * (unsigned)chunk);
        }
    }
    {
        const int p = w;
        if (BS) {
            const int krow = 4 * p + (lane >> 4), slot = lane & 15;
            const int chunk = slot ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
            vb = 2u * ((unsigned)krow * ldb + 8u * (unsigned)chunk);
        } else {
            const int row = 16 * p + (lane >> 2), slot = lane & 3;
            const int chunk = slot ^ (((row >> 3) & 1) << 1);
            vb = 2u * ((unsigned)row * ldb + 8u * (unsigned)chunk);
        }
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // S image of A: piece p -> block p >> 3 (8 KB), k-rows 4 (p & 7) .. + 3 (1 KB); T image: rows 16 p .. + 15 (1 KB)
    const unsigned m0a0 = lds0 + (AS ? 8192u * (unsigned)((2 * w) >> 3) + 1024u * (unsigned)((2 * w) & 7) : 1024u * (unsigned)(2 * w));
    const unsigned m0a1 = lds0 + (AS ? 8192u * (unsigned)((2 * w + 1) >> 3) + 1024u * (unsigned)((2 * w + 1) & 7)
                                      : 1024u * (unsigned)(2 * w + 1));
    const unsigned m0b = lds0 + B_A_BYTES + 1024u * (unsigned)w;
#define G16B_ISSUE(stage)                                              \
    {                                                                  \
        const unsigned so = (unsigned)(stage) * (unsigned)B_ST_BYTES;  \
        dma16(sa, va[0], m0a0 + so);                                   \
        dma16(sa, va[1], m0a1 + so);                                   \
        dma16(sb, vb, m0b + so);                                       \
        sa += sa_step;                                                 \
        sb += sb_step;                                                 \
    }
#define G16B_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")

    const int l15 = lane & 15, g4 = lane >> 4;
    // T images: block i (16 rows): row r0 + 16 i + l15, chunk g4 (swizzled by bit 3 of the row = bit 3 of l15)
    const int fa = (64 * wm + l15) * 64 + ((g4 ^ (((l15 >> 3) & 1) << 1)) << 4);
    const int fb = B_A_BYTES + (64 * wn + l15) * 64 + ((g4 ^ (((l15 >> 3) & 1) << 1)) << 4);
    // S images: k-row 8 g4 + 4 h2 + q, q = l15 >> 2; columns 4 (l15 & 3) .. of the block (gemm16.h, k32 step 0)
    const int q = l15 >> 2, pp = l15 & 3;
    int sbase[2], ssw[2];  // per k-row half h2: 256 krow + 8 (pp & 1), and the swizzle of that k-row
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int krow = 8 * g4 + 4 * h2 + q;
        ssw[h2] = ((krow & 3) << 2) | ((krow >> 2) & 3);
        sbase[h2] = 256 * krow + 8 * (pp & 1);
    }

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define G16B_TR(addr) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(addr))

    G16B_ISSUE(0);
    if (nk > 1) G16B_ISSUE(1);
    if (w >= 4) __builtin_amdgcn_s_setprio(1);
    for (int t = 0; t < nk; ++t) {
        // stage t has landed (the DMA of t + 1 may still fly) and every wave is done with stage t - 1
        if (t + 1 < nk) G16B_WAIT_BARRIER(B_NDMA);
        else G16B_WAIT_BARRIER(0);
        if (t + 2 < nk) G16B_ISSUE((t + 2) % B_NST);
        const char* st = lds + (t % B_NST) * B_ST_BYTES;
        bf16x8 fa_[4], fb_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (AS) {
                const int ca = ((64 * (wm & 1) + 16 * i) >> 3) + (pp >> 1);
                const char* b0 = st + 8192 * (wm >> 1);
                const s16x4 lo = G16B_TR(b0 + sbase[0] + 16 * (ca ^ ssw[0])), hi = G16B_TR(b0 + sbase[1] + 16 * (ca ^ ssw[1]));
                fa_[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                fa_[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + fa + 1024 * i));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BS) {
                const int cb = ((64 * wn + 16 * j) >> 3) + (pp >> 1);
                const char* b0 = st + B_A_BYTES;
                const s16x4 lo = G16B_TR(b0 + sbase[0] + 16 * (cb ^ ssw[0])), hi = G16B_TR(b0 + sbase[1] + 16 * (cb ^ ssw[1]));
                fb_[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                fb_[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(st + fb + 1024 * j));
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = mfma16<F16>(fb_[j], fa_[i], acc[i][j]);
    }
#undef G16B_ISSUE
#undef G16B_WAIT_BARRIER
#undef G16B_TR

    // ---- epilogue (gemm16.h's: the tile through LDS, whole rows out as 16-byte write-through stores); the float32 tile
    // (132 KB) goes in two passes of 128 rows through the 72 KB of the ring
    constexpr int RS = O16 ? 272 : 528;
    constexpr int NPASS = O16 ? 1 : 2;
    constexpr int ROWS = BM / NPASS;
    float ss = 0.f;
    char* cbase = reinterpret_cast<char*>(P.C) + ((long)slice * a.slice_stride + (long)BM * tm * P.ldc + BN * tn) * (O16 ? 2 : 4);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        __syncthreads();  // every wave is done with the last stage / with the previous pass's rows
        if (NPASS == 1 || (wm >> 1) == ps) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.bias) bv = *reinterpret_cast<const float4*>(P.bias + BN * tn + 64 * wn + 4 * g4 + 16 * j);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float c0 = acc[i][j][0] + bv.x, c1 = acc[i][j][1] + bv.y, c2 = acc[i][j][2] + bv.z,
                                c3 = acc[i][j][3] + bv.w;
                    ss = fmaf(c0, c0, ss); ss = fmaf(c1, c1, ss); ss = fmaf(c2, c2, ss); ss = fmaf(c3, c3, ss);
                    char* dst = lds + (64 * wm + 16 * i + l15 - ROWS * ps) * RS + (64 * wn + 16 * j + 4 * g4) * (O16 ? 2 : 4);
                    if (O16) *reinterpret_cast<uint2*>(dst) = make_uint2(pack_h<F16>(c0, c1), pack_h<F16>(c2, c3));
                    else *reinterpret_cast<float4*>(dst) = make_float4(c0, c1, c2, c3);
                }
            }
        }
        __syncthreads();
        constexpr int LPR = O16 ? 16 : 32;
        constexpr int RPP = 512 / LPR;
        const int rr = tid / LPR, cc = tid % LPR;
        for (int r = rr; r < ROWS; r += RPP) {
            const uint4 v = *reinterpret_cast<const uint4*>(lds + r * RS + 16 * cc);
            char* dstp = cbase + (long)(r + ROWS * ps) * P.ldc * (O16 ? 2 : 4) + 16 * cc;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 vv = {v.x, v.y, v.z, v.w};
            // (s_nop 1 inside the string: hipcc pads nothing behind an asm store - its next instruction may overwrite the
            // data registers before the store has read them: intermittent wrong elements, found in round 6)
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstp), "v"(vv) : "memory");
        }
    }
    if (P.sumsq) {
        ss = nsvd_wave_sum(ss);
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        if (lane == 0) red[w] = ss;
        __syncthreads();
        if (tid == 0)
            P.sumsq[slice * per + tm * P.tiles_n + tn] =
                ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
}

// launch the two-workgroups-per-CU form (arguments already validated and completed by launch(): tiles, wg0, nwg)
template <bool AS_, bool BS_, bool O_, bool F_>
inline int launch_b_inst(const Args& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm16b_kernel<AS_, BS_, O_, F_>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm16b_kernel<AS_, BS_, O_, F_>), dim3((unsigned)a.nwg), dim3(512), B_LDS_BYTES, s, a);
    return 0;
}

}  // namespace nsvd_g16
