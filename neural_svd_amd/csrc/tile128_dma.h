// C (128 x 128) += A (128 rows) . B (128 rows)^T over a run of 32-wide contraction chunks, both operands contraction-
// contiguous ("NT"), fp32-input MFMA, operands staged global -> LDS by LDS-DMA (global_load_lds_dwordx4): no staging
// registers, no ds_write, and a prefetch distance a register-staged loop cannot have. The K loop shared by the
// layer-0 weight gradient (pmlp_bwd.hip) and the tower GEMMs (tower.hip).
// 256 threads = 4 waves as 2 x 2 of 64 x 64 (wave (wm, wn) owns rows 64 wm of A x rows 64 wn of B: acc[i][j] = its
// 32 x 32 block (32 i, 32 j)).
//   a_base / b_base: row 0, chunk 0 of the tile's operand rows (workgroup-uniform); a_ld / b_ld: floats between rows
//   (a tile spans less than 4 GB); chunk c starts at column 32 c;  lds: T128D_LDS_FLOATS floats (64 KB).
//
// Why (measured on MI355X, 16 chunks of the layer-0 weight gradient at cfg2, cycles per tile loop, alone on the chip /
// next to the co-resident small workgroups of that kernel; 65.5 K is the MFMA issue time): the register-staged
// loop this replaces (global -> 8 float4 per thread -> ds_write -> padded LDS tiles, one barrier per chunk) loads a
// staging register 0.75 of a chunk (~3.8 K cycles) before its ds_write - less than the loaded memory latency - and
// waits on memory in every chunk: 81 K / 90 K. Refilling a register right behind its own ds_write (a full chunk of
// lead) is worse, 107 K / 114 K: the load waits for the LDS unit to have read the register it overwrites. A second
// register set costs 32 VGPRs the kernel does not have at two workgroups per CU (the compiler spilled). This loop:
// 69.7 K / 73.4 K, and 32 VGPRs fewer, which the weight-gradient kernel spends on prefetching optimiser state.
//
// Layout: a ring of four stages, each one HALF chunk (16 columns) of both operands: [A: 128 rows x 64 B | B: 128 rows
// x 64 B] = 16 KB, unpadded. One DMA instruction moves 16 rows (lane i -> row i / 4, 16-B slot i % 4); the slot of
// row r holds column quad slot ^ ((r >> 2) & 3) - the swizzle is applied on the SOURCE address and again on the
// fragment reads, which makes the ds_read_b128 of 16 consecutive rows hit 16 distinct 4-bank groups.
// Pipeline, per half chunk h (stage h & 3): first q-group = 16 MFMAs on the fragments read one q-group earlier +
// the reads of its second q-group; then "the DMA of half chunk h + 1 has landed" (s_waitcnt vmcnt with the two
// younger half chunks still in flight) and the workgroup barrier - which also says every wave is done reading stage
// h & 3; second q-group = 16 MFMAs + the first fragments of half chunk h + 1 + the DMA of half chunk h + 4 into the
// stage just freed. A DMA has 1.5 chunks to land.
// The DMA instructions and their waits are inline asm: the compiler, knowing of an LDS-DMA in flight, puts
// s_waitcnt vmcnt(0) in front of EVERY LDS read. Unknown to it, they only make its own vmcnt waits conservative
// (the counter retires in order).
//
// Hook: a caller may hang global loads for its epilogue on the last four half chunks: hook.issue<k, part>(),
// k = 0..7, part = 0..3, is called behind the part-th MFMA block of the (k & 1)-th q-group of the (k / 2)-th of
// them; every call must issue exactly Hook::LOADS vector memory loads (the DMA waits count them). Called only when
// the pipeline runs (nch >= 4): the return value. Tile128NoHook issues nothing.
#pragma once
#include "pmlp_common.h"

namespace nsvd_pmlp {

constexpr int T128D_LDS_FLOATS = 4 * 4096;  // four stages of 16 KB

struct Tile128NoHook {
    static constexpr int LOADS = 0;
    template <int K, int PART>
    __device__ __forceinline__ void issue() {}
};

// 16 B per lane global -> LDS: source = sbase (uniform) + voff (per lane, bytes), destination = m0 (wave-uniform LDS
// byte address) + lane * 16. M0 is a reserved register (never allocated; hipcc rejects it on a clobber list): the
// compiler writes it itself right in front of each of its own uses, and the kernels that include this header have
// none (checked in their ISA) - do not mix this routine with the LDS-DMA builtin in one kernel.
__device__ __forceinline__ void t128d_dma(const float* sbase, unsigned voff, unsigned m0) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0), "v"(voff), "s"(sbase) : "memory");
}

// NJ = 2: the 128 x 128 tile; NJ = 1: 128 rows of A x 64 rows of B (wave (wm, wn): rows 64 wm of A x rows 32 wn of B,
// acc[i][0] = its block (32 i, 0)) - twice the tiles for outputs that would otherwise leave half the CUs without one
// BF = true: the operands are bfloat16 (the pointers, leading dimensions and chunk counts stay in FLOAT units: a row
// of K bf16 values is K / 2 floats, a chunk 64 of them). The bytes move exactly as for float32 - same DMA, same
// swizzle, same 16-byte fragment reads - and a fragment (8 bf16 per lane, lanes 0-31 / 32-63 the two halves of 16
// contraction indices) feeds ONE v_mfma_f32_32x32x16_bf16 where four float32 fragments' components fed four
// v_mfma_f32_32x32x2_f32: the loop turns from MFMA-bound into staging-bound (mixed-precision tower GEMMs, tower.hip).
typedef __bf16 nsvd_bf16x8 __attribute__((ext_vector_type(8)));
// GA = true: the rows of A are GATHERED - ga0 / ga1 = byte offsets from a_base of this lane's two rows (tile rows
// 32 w + lane / 4 and + 16; w = wave), a_ld unused (the dense kernel operator's K[rows, :], kernel_apply.hip).
template <class Hook, int NJ, bool BF = false, bool GA = false>
__device__ __forceinline__ bool nsvd_tile128_dma(const float* a_base, const float* b_base, unsigned a_ld, unsigned b_ld,
                                                 int nch, float* lds, f32x16 (&acc)[2][NJ], Hook& hook,
                                                 unsigned ga0 = 0, unsigned ga1 = 0) {
    static_assert(NJ == 1 || NJ == 2, "B operand: 64 or 128 rows");
    const int tid = threadIdx.x & 255;  // four waves; a larger workgroup may run the routine in one of its wave groups
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    // DMA: this wave moves rows 32 w .. 32 w + 31 of A and 16 NJ w .. + 16 NJ - 1 of B, 16 rows per instruction
    const unsigned dq = 4u * (unsigned)((lane & 3) ^ ((lane >> 4) & 3));  // source column (floats) of this lane's slot
    const unsigned va0 = GA ? ga0 + 4u * dq : 4u * ((unsigned)(32 * w + (lane >> 2)) * a_ld + dq);
    const unsigned va1 = GA ? ga1 + 4u * dq : va0 + 64u * a_ld;
    const unsigned vb0 = 4u * ((unsigned)(16 * NJ * w + (lane >> 2)) * b_ld + dq), vb1 = vb0 + 64u * b_ld;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned m0a = lds0 + 2048u * (unsigned)w, m0b = lds0 + 8192u + 1024u * NJ * (unsigned)w;
    // fragments: row 64 wm + li (+ 32) of A, 32 NJ wn + li (+ 32) of B; column quad 2 q + hi of the half chunk
    const int fz = (li >> 2) & 3;
    const char* ldsb = reinterpret_cast<const char*>(lds);
    const char* fa0 = ldsb + (64 * wm + li) * 64 + ((hi ^ fz) << 4);
    const char* fa1 = ldsb + (64 * wm + li) * 64 + (((2 + hi) ^ fz) << 4);
    const char* fb0 = ldsb + 8192 + (32 * NJ * wn + li) * 64 + ((hi ^ fz) << 4);
    const char* fb1 = ldsb + 8192 + (32 * NJ * wn + li) * 64 + (((2 + hi) ^ fz) << 4);
    struct F4 {
        float4 a0, a1, b0, b1;
    };
    F4 f0, f1;
#define TD_DMA(s, pa, pb)                                          \
    {                                                              \
        t128d_dma(pa, va0, m0a + (s) * 16384u);                    \
        t128d_dma(pa, va1, m0a + (s) * 16384u + 1024u);            \
        t128d_dma(pb, vb0, m0b + (s) * 16384u);                    \
        if (NJ == 2) t128d_dma(pb, vb1, m0b + (s) * 16384u + 1024u); \
    }
#define TD_RD(p, s) (*reinterpret_cast<const float4*>((p) + (s) * 16384))
#define TD_RDB1(p, s) (NJ == 2 ? TD_RD((p) + 2048, s) : make_float4(0.f, 0.f, 0.f, 0.f))
#define TD_BFR(v) __builtin_bit_cast(nsvd_bf16x8, v)
#define TD_FIRST_x 1
#define TD_FIRST_y 0
#define TD_FIRST_z 0
#define TD_FIRST_w 0
// float32: component X of the four fragments; bfloat16: the whole fragments, once (at X = x)
#define TD_MMA4(f, X)                                                                                   \
    if (!BF) {                                                                                          \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b0.X, acc[0][0], 0, 0, 0);          \
        if (NJ == 2) acc[0][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b1.X, acc[0][NJ - 1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b0.X, acc[1][0], 0, 0, 0);          \
        if (NJ == 2) acc[1][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b1.X, acc[1][NJ - 1], 0, 0, 0); \
    } else if (TD_FIRST_##X) {                                                                          \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TD_BFR(f.a0), TD_BFR(f.b0), acc[0][0], 0, 0, 0); \
        if (NJ == 2) acc[0][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TD_BFR(f.a0), TD_BFR(f.b1), acc[0][NJ - 1], 0, 0, 0); \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TD_BFR(f.a1), TD_BFR(f.b0), acc[1][0], 0, 0, 0); \
        if (NJ == 2) acc[1][NJ - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(TD_BFR(f.a1), TD_BFR(f.b1), acc[1][NJ - 1], 0, 0, 0); \
    }
#define TD_FENCE() __builtin_amdgcn_sched_barrier(0)
#define TD_WAIT_BARRIER(n) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(n) : "memory")
#define TD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    const int n = nch >= 4 ? (nch & ~1) : 0;  // chunks the pipeline takes (even), after c0 = nch - n plain ones
    const int c0 = nch - n;
    const float* pa = a_base;
    const float* pb = b_base;
    for (int c = 0; c < c0; ++c) {
        // a chunk on its own: both halves in, then its four q-groups
        TD_BARRIER();  // the previous chunk's fragment reads are done
        TD_DMA(0, pa, pb);
        TD_DMA(1, pa + 16, pb + 16);
        pa += 32;
        pb += 32;
        TD_WAIT_BARRIER(0);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f0.a0 = TD_RD(fa0, s); f0.a1 = TD_RD(fa0 + 2048, s); f0.b0 = TD_RD(fb0, s); f0.b1 = TD_RDB1(fb0, s);
            f1.a0 = TD_RD(fa1, s); f1.a1 = TD_RD(fa1 + 2048, s); f1.b0 = TD_RD(fb1, s); f1.b1 = TD_RDB1(fb1, s);
            TD_MMA4(f0, x) TD_MMA4(f0, y) TD_MMA4(f0, z) TD_MMA4(f0, w)
            TD_MMA4(f1, x) TD_MMA4(f1, y) TD_MMA4(f1, z) TD_MMA4(f1, w)
        }
    }
    if (n == 0) return false;

    // half chunk on stage s: DO_DMA refills the stage with the half chunk four ahead; WAITN = vector memory operations
    // that may still be in flight when the next half chunk must have landed; K0 / K1 = hook slots (-1: none);
    // LAST: no next half chunk
#define TD_HALF(s, DO_DMA, WAITN, K0, K1, LAST)                                                           \
    {                                                                                                     \
        TD_MMA4(f0, x) f1.a0 = TD_RD(fa1, s);                                                             \
        if (K0 >= 0) hook.template issue<(K0 >= 0 ? K0 : 0), 0>();                                        \
        TD_FENCE();                                                                                       \
        TD_MMA4(f0, y) f1.a1 = TD_RD(fa1 + 2048, s);                                                      \
        if (K0 >= 0) hook.template issue<(K0 >= 0 ? K0 : 0), 1>();                                        \
        TD_FENCE();                                                                                       \
        TD_MMA4(f0, z) f1.b0 = TD_RD(fb1, s);                                                             \
        if (K0 >= 0) hook.template issue<(K0 >= 0 ? K0 : 0), 2>();                                        \
        TD_FENCE();                                                                                       \
        TD_MMA4(f0, w) f1.b1 = TD_RDB1(fb1, s);                                                           \
        if (K0 >= 0) hook.template issue<(K0 >= 0 ? K0 : 0), 3>();                                        \
        TD_FENCE();                                                                                       \
        if (!(LAST)) TD_WAIT_BARRIER((WAITN) > 63 ? 63 : (WAITN));                                        \
        TD_MMA4(f1, x) if (!(LAST)) f0.a0 = TD_RD(fa0, ((s) + 1) & 3);                                    \
        if (DO_DMA) t128d_dma(pa, va0, m0a + (s) * 16384u);                                               \
        if (K1 >= 0) hook.template issue<(K1 >= 0 ? K1 : 0), 0>();                                        \
        TD_FENCE();                                                                                       \
        TD_MMA4(f1, y) if (!(LAST)) f0.a1 = TD_RD(fa0 + 2048, ((s) + 1) & 3);                             \
        if (DO_DMA) t128d_dma(pa, va1, m0a + (s) * 16384u + 1024u);                                       \
        if (K1 >= 0) hook.template issue<(K1 >= 0 ? K1 : 0), 1>();                                        \
        TD_FENCE();                                                                                       \
        TD_MMA4(f1, z) if (!(LAST)) f0.b0 = TD_RD(fb0, ((s) + 1) & 3);                                    \
        if (DO_DMA) t128d_dma(pb, vb0, m0b + (s) * 16384u);                                               \
        if (K1 >= 0) hook.template issue<(K1 >= 0 ? K1 : 0), 2>();                                        \
        TD_FENCE();                                                                                       \
        TD_MMA4(f1, w) if (!(LAST)) f0.b1 = TD_RDB1(fb0, ((s) + 1) & 3);                                  \
        if (DO_DMA) {                                                                                     \
            if (NJ == 2) t128d_dma(pb, vb1, m0b + (s) * 16384u + 1024u);                                  \
            pa += 16;                                                                                     \
            pb += 16;                                                                                     \
        }                                                                                                 \
        if (K1 >= 0) hook.template issue<(K1 >= 0 ? K1 : 0), 3>();                                        \
        TD_FENCE();                                                                                       \
    }
    constexpr int HL = Hook::LOADS;
    constexpr int G = 2 + NJ;  // DMA instructions per wave and half chunk
    if (c0) TD_BARRIER();
    TD_DMA(0, pa, pb);
    TD_DMA(1, pa + 16, pb + 16);
    TD_DMA(2, pa + 32, pb + 32);
    TD_DMA(3, pa + 48, pb + 48);
    pa += 64;
    pb += 64;
    TD_WAIT_BARRIER(3 * G);
    f0.a0 = TD_RD(fa0, 0); f0.a1 = TD_RD(fa0 + 2048, 0); f0.b0 = TD_RD(fb0, 0); f0.b1 = TD_RDB1(fb0, 0);
    for (int g = n / 2 - 1; g > 0; --g) {
        TD_HALF(0, true, 2 * G, -1, -1, false)
        TD_HALF(1, true, 2 * G, -1, -1, false)
        TD_HALF(2, true, 2 * G, -1, -1, false)
        TD_HALF(3, true, 2 * G, -1, -1, false)
    }
    // the last four half chunks: nothing left to fetch; the hook's loads are younger than every DMA
    TD_HALF(0, false, 2 * G + 4 * HL, 0, 1, false)
    TD_HALF(1, false, G + 12 * HL, 2, 3, false)
    TD_HALF(2, false, 20 * HL, 4, 5, false)
    TD_HALF(3, false, 0, 6, 7, true)
#undef TD_HALF
#undef TD_DMA
#undef TD_RD
#undef TD_RDB1
#undef TD_MMA4
#undef TD_BFR
#undef TD_FIRST_x
#undef TD_FIRST_y
#undef TD_FIRST_z
#undef TD_FIRST_w
#undef TD_FENCE
#undef TD_WAIT_BARRIER
#undef TD_BARRIER
    return true;
}

template <int NJ, bool BF = false>
__device__ __forceinline__ void nsvd_tile128_dma(const float* a_base, const float* b_base, unsigned a_ld, unsigned b_ld,
                                                 int nch, float* lds, f32x16 (&acc)[2][NJ]) {
    Tile128NoHook none;
    nsvd_tile128_dma<Tile128NoHook, NJ, BF>(a_base, b_base, a_ld, b_ld, nch, lds, acc, none);
}

}  // namespace nsvd_pmlp
