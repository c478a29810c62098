// pmlp_wgrad_pipe_kernel: the weight-gradient kernel as a persistent, software-pipelined workgroup per CU.
// Included by pmlp_bwd.hip (inside its anonymous namespace, after WgradArgs / wg_emit1).
//
// Why: in pmlp_fused_wgrad_kernel the dW_i quadrants co-reside with the dW_0 tiles and stretch their K loops from 72 K
// to 94 K cycles, and the optimiser epilogue of the 256 dW_0 tiles (24 B/parameter: read p, sq, ema, write them back)
// then runs with the matrix pipes idle. Here a workgroup is 8 waves and every CU walks the same short item list - its
// 128 x 128 dW_0 tile, then one 64 x 32 piece of a hidden-layer gradient dW_i (contraction split in two halves over the
// wave pairs, so that all 256 CUs share that work evenly):
//   waves 0-3  "MFMA waves": fragment reads and MFMAs only; after an item's K loop they drop the accumulators into an
//              LDS hand-off tile and go on with the next item;
//   waves 4-7  "staging / epilogue waves": (a) ALL the staging - the global -> register -> LDS copies of every chunk of
//              the item list as one continuous stream, two chunks ahead of the multiplications (an item's first chunk
//              is in LDS before the previous item has been handed off); (b) the PREVIOUS item's epilogue from the
//              hand-off tile - gradient store and / or RMSprop + EMA step - while the MFMA waves are in the next K loop:
//              the optimiser traffic of the dW_0 tile flies under the dW_i piece; (c) the bias gradients (row sums of
//              what they stage) and, at the end, the 128 -> 1 layer, db_last and d scales.
// Staged bytes per CU are what bounds this kernel's K loops (L2 / Infinity-Cache -> CU at ~8-12 B/clk/CU), which is why
// the dW_0 tile is NOT cut into halves that would each re-stage dz_0 (tried: 768 KB instead of 512 KB per CU, K loops
// 1.5x slower, although the epilogue then hides completely).
// s_barrier is workgroup-wide on gfx950, so the staging waves execute exactly the barriers of the K loop they
// feed (one per 32-row chunk) and do one slot of epilogue work between two of them: state loads are issued one
// slot before they are consumed. Correctness depends only on the barrier COUNTS agreeing (both sides derive them from
// the same item descriptor), never on timing.
// Split-K launches (head-parallel ranks, S > 1) keep the tile kernel above.
#pragma once

constexpr int PIPE_THREADS = 512;
constexpr int PIPE_KC = BK;                            // contraction chunk: 32 rows of the batch per barrier
constexpr int PIPE_LD = A_LD;                          // padded stage row (36 floats): conflict-free ds_read_b128
constexpr int PIPE_HS_LD = HID + 8;                    // hand-off tile row (floats)
constexpr int PIPE_SROWS = 256;                        // staged rows per chunk: two 128-row regions
constexpr int PIPE_SBUF = PIPE_SROWS * PIPE_LD;        // one stage buffer (floats)
constexpr int PIPE_HAND = HID * PIPE_HS_LD;            // the hand-off tile (floats)
constexpr int PIPE_LDS_FLOATS = 2 * PIPE_SBUF + PIPE_HAND;
constexpr size_t PIPE_LDS_BYTES = (size_t)PIPE_LDS_FLOATS * sizeof(float);  // 143 360 B: one workgroup per CU

#ifdef NSVD_WG_STAMPS
// diagnostic build: per workgroup, cycle stamps of the MFMA side (thread 0) and of the epilogue side (thread 256)
__device__ unsigned long long g_pipe_stamps[1024 * 32];
#define PIPE_STAMP(slot) g_pipe_stamps[(size_t)blockIdx.x * 32 + (slot)] = __builtin_readcyclecounter()
#define PIPE_STAMP_M(slot) if (threadIdx.x == 0) PIPE_STAMP(slot)
__device__ unsigned long long g_pipe_chunk_stamps[1024 * 64];  // [block][item (<4)][chunk (<16)]
__device__ int g_pipe_item;                                    // unused; keeps the symbol table simple
#define PIPE_STAMP_CH(kk, c)                                                                     \
    if (threadIdx.x == 0 && (kk) < 4 && (c) < 16)                                                \
        g_pipe_chunk_stamps[(size_t)blockIdx.x * 64 + (kk) * 16 + (c)] = __builtin_readcyclecounter()
#define PIPE_STAMP_E(slot) if (threadIdx.x == 256) PIPE_STAMP(slot)
#else
#define PIPE_STAMP_M(slot)
#define PIPE_STAMP_E(slot)
#define PIPE_STAMP_CH(kk, c)
#endif
#ifdef NSVD_WG_STAMPS
// staging-side stamps of steps 2..5 (four per step) in row 3 of the chunk-stamp table
#define PIPE_STAMP_H(gg, i)                                                                        \
    if (threadIdx.x == 256 && (gg) >= 2 && (gg) < 6)                                               \
        g_pipe_chunk_stamps[(size_t)blockIdx.x * 64 + 48 + 4 * ((gg) - 2) + (i)] = __builtin_readcyclecounter()
#else
#define PIPE_STAMP_H(gg, i)
#endif

struct PipeItem {
    int kind;   // 0: a dW_0 tile (128 rows x 128 feature columns, K = B);  1: dW_i piece (64 x 32, K = 2 x B/2)
    int l, i;   // head, layer
    int n0, k0; // origin inside W_i[l]: rows n0.., columns k0..
    int ldw;    // row length of W_i[l]
    int nch;    // 32-row chunks each wave contracts over (one barrier per chunk)
    int bias;   // the item also carries the row sums of its dz rows (the bias gradient)
};

__device__ __forceinline__ int pipe_xcd_remap(int idx, int n) {
    // consecutive indices (same head / layer) on the same XCD when the count allows it: blockIdx % 8 is the XCD
    return (n & 7) == 0 ? (idx & 7) * (n >> 3) + (idx >> 3) : idx;
}

__device__ __forceinline__ int pipe_num_items(const WgradArgs& a) { return a.nA + 8 * (a.nlayers - 2) * a.L; }

__device__ __forceinline__ PipeItem pipe_decode(const WgradArgs& a, int it) {
    PipeItem p;
    const int nA = a.nA;
    if (it < nA) {
        const int unit = pipe_xcd_remap(it, nA);
        const int nkt = a.F / HID;
        p.kind = 0;
        p.l = unit / nkt;
        p.i = 0;
        p.n0 = 0;
        p.k0 = (unit - p.l * nkt) * HID;
        p.ldw = a.F;
        p.nch = a.B / PIPE_KC;
        p.bias = p.k0 == 0;
        return p;
    }
    const int nP = 8 * (a.nlayers - 2) * a.L;  // 8 pieces per (layer, head)
    const int q = pipe_xcd_remap(it - nA, nP);
    const int piece = q & 7, rest = q >> 3;
    p.kind = 1;
    p.l = rest % a.L;
    p.i = 1 + rest / a.L;
    p.n0 = 64 * (piece >> 2);
    p.k0 = 32 * (piece & 3);
    p.ldw = HID;
    p.nch = a.B / (2 * PIPE_KC);
    p.bias = p.k0 == 0;
    return p;
}

// ---- stage layout ------------------------------------------------------------------------------------------------
// One chunk = 32 rows of the batch; its stage image is 256 rows of PIPE_LD = 36 floats:
//   kind 0: rows 0..127 = dz_0[l][n][chunk], rows 128..255 = phi^T[k0 + r][chunk]
//   kind 1: rows 0..63 / 64..127 = dz_i[l][n0 + r][chunk of batch half 0 / 1], rows 128..159 / 160..191 =
//           a_{i-1}[l][k0 + r][chunk of batch half 0 / 1] (rows 192..255 unused)
// Staging thread t (of the 256 of the staging waves) moves the float4 (row (t >> 3) + 32 j, columns 4 (t & 7)..) of
// pass j = 0..7 (passes 0-3: the first region, 4-7: the second).
struct PipeSrc {
    const float* pa;   // row (t >> 3) of the first region, this thread's columns of chunk 0
    const float* pb;   // row (t >> 3) of the second region
};

__device__ __forceinline__ PipeSrc pipe_sources(const WgradArgs& a, const PipeItem& p, int ht) {
    PipeSrc s;
    const int s_row = ht >> 3, s_c4 = ht & 7;
    const size_t B = (size_t)a.B;
    if (p.kind == 0) {
        s.pa = a.dz[0] + ((size_t)p.l * HID + s_row) * B + 4 * s_c4;
        s.pb = a.phiTc + ((size_t)p.k0 + s_row) * B + 4 * s_c4;
    } else {
        s.pa = a.dz[p.i] + ((size_t)p.l * HID + p.n0 + s_row) * B + 4 * s_c4;
        s.pb = a.zsave[p.i - 1] + ((size_t)p.l * HID + p.k0 + s_row) * B + 4 * s_c4;
    }
    return s;
}

// element offset of staging pass j relative to pa (j < 4) / pb (j >= 4); kind 1 has no passes 6, 7 (they re-read
// pass 4 / 5: the load count per chunk stays fixed, which keeps the compiler's vmcnt bookkeeping exact)
__device__ __forceinline__ size_t pipe_pass_off(int kind, int j, size_t B) {
    if (kind == 0) return (size_t)(32 * (j & 3)) * B;
    if (j < 4) return (size_t)(32 * (j & 1)) * B + (size_t)(j >> 1) * (B / 2);
    return (size_t)(j & 1) * (B / 2);
}

// ---- MFMA waves ------------------------------------------------------------------------------------------------
struct PipeFrag {
    float4 a0, a1, b0, b1;
};

// The nch chunks of one item: fragment reads and MFMAs only (the staging waves keep the two stage buffers filled).
// g = this workgroup's running chunk counter (buffer = g & 1). One barrier per chunk: "chunk g is in LDS"; it sits
// before the LAST q-group of the previous chunk, whose MFMAs cover the first fragment reads of the new one.
template <int KIND>
__device__ __forceinline__ void pipe_mfma_item(const float* stage, int& g, int nch, f32x16 (&acc)[4], int tid, int kk) {
    (void)kk;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    // kind 0: wave (wm, wn) = 64 x 64 of the tile; kind 1: wave (wm, kh) = 32 x 32 of the piece over batch half kh
    const int arow = KIND == 0 ? 64 * (w >> 1) + li : 64 * (w >> 1) + 32 * (w & 1) + li;
    const int brow = KIND == 0 ? 128 + 64 * (w & 1) + li : 128 + 32 * (w >> 1) + li;
    const float* fa = stage + arow * PIPE_LD + 4 * hi;
    const float* fb = stage + brow * PIPE_LD + 4 * hi;
#define PK_READ(f, buf, q)                                                                                 \
    {                                                                                                      \
        f.a0 = *reinterpret_cast<const float4*>(fa + (buf) * PIPE_SBUF + 8 * (q));                         \
        f.b0 = *reinterpret_cast<const float4*>(fb + (buf) * PIPE_SBUF + 8 * (q));                         \
        if (KIND == 0) {                                                                                   \
            f.a1 = *reinterpret_cast<const float4*>(fa + (buf) * PIPE_SBUF + 32 * PIPE_LD + 8 * (q));      \
            f.b1 = *reinterpret_cast<const float4*>(fb + (buf) * PIPE_SBUF + 32 * PIPE_LD + 8 * (q));      \
        }                                                                                                  \
    }
#define PK_MMA1(f, X)                                                                                      \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b0.X, acc[0], 0, 0, 0);                        \
    if (KIND == 0) {                                                                                       \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b1.X, acc[1], 0, 0, 0);                    \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b0.X, acc[2], 0, 0, 0);                    \
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b1.X, acc[3], 0, 0, 0);                    \
    }
#define PK_MMA(f) PK_MMA1(f, x) PK_MMA1(f, y) PK_MMA1(f, z) PK_MMA1(f, w)
#define PK_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PK_IL(n, mask)                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier((mask), 1, 0);        \
    }
#define PK_NR (KIND == 0 ? 4 : 2)
#define PK_STEP(fr, fm, buf, q) \
    {                           \
        PK_READ(fr, buf, q);    \
        PK_MMA(fm);             \
        PK_IL(PK_NR, 0x100);    \
        PK_FENCE();             \
    }
    PipeFrag f0, f1;
    f0.a1 = f1.a1 = f0.b1 = f1.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk g is staged
    PIPE_STAMP_CH(kk, 0);
    PK_READ(f0, g & 1, 0);
    // all chunks but the last: the next chunk's barrier and first fragment reads sit before the last q-group
    for (int c = 0; c + 1 < nch; ++c, ++g) {
        const int cur = g & 1;
        PK_STEP(f1, f0, cur, 1);
        PK_STEP(f0, f1, cur, 2);
        PK_STEP(f1, f0, cur, 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk g + 1 is staged
        PIPE_STAMP_CH(kk, c + 1);
        PK_READ(f0, cur ^ 1, 0);
        PK_MMA(f1);
        PK_IL(PK_NR, 0x100);
        PK_FENCE();
    }
    {
        const int cur = g & 1;
        PK_STEP(f1, f0, cur, 1);
        PK_STEP(f0, f1, cur, 2);
        PK_STEP(f1, f0, cur, 3);
        PK_MMA(f1);
        ++g;
    }
#undef PK_STEP
#undef PK_NR
#undef PK_IL
#undef PK_FENCE
#undef PK_MMA
#undef PK_MMA1
#undef PK_READ
}

// accumulators -> hand-off tile
template <int KIND>
__device__ __forceinline__ void pipe_handoff(const f32x16 (&acc)[4], float* hand, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    if (KIND == 0) {
        const int wm = w >> 1, wn = w & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    hand[(64 * wm + 32 * i + acc_row(r, hi)) * PIPE_HS_LD + 64 * wn + 32 * j + li] = acc[2 * i + j][r];
    } else {
        const int wm = w & 1, kh = w >> 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) hand[(64 * kh + 32 * wm + acc_row(r, hi)) * PIPE_HS_LD + li] = acc[0][r];
    }
}

// ---- staging (done by the epilogue waves) ------------------------------------------------------------------------
// A cursor over this workgroup's chunk stream: item after item, chunk after chunk.
struct PipeCursor {
    int it;        // item index (>= nItems: exhausted)
    int c;         // chunk inside the item
    int nch, kind;
    PipeSrc src;
};

__device__ __forceinline__ void pipe_cursor_open(PipeCursor& cu, const WgradArgs& a, int it, int nItems, int ht) {
    cu.it = it;
    cu.c = 0;
    if (it < nItems) {
        const PipeItem p = pipe_decode(a, it);
        cu.nch = p.nch;
        cu.kind = p.kind;
        cu.src = pipe_sources(a, p, ht);
    } else {
        cu.nch = 0;
        cu.kind = 0;
        cu.src.pa = cu.src.pb = nullptr;
    }
}

__device__ __forceinline__ void pipe_cursor_next(PipeCursor& cu, const WgradArgs& a, int nItems, int stride, int ht) {
    if (++cu.c >= cu.nch) pipe_cursor_open(cu, a, cu.it + stride, nItems, ht);
}

// the eight staging registers of one chunk
struct PipeRegs {
    float4 r0, r1, r2, r3, r4, r5, r6, r7;
};

__device__ __forceinline__ void pipe_regs_load(PipeRegs& R, const PipeCursor& cu, size_t B) {
    const float* pa = cu.src.pa + (size_t)cu.c * PIPE_KC;
    const float* pb = cu.src.pb + (size_t)cu.c * PIPE_KC;
    const int k = cu.kind;
#define PL(R_, j) R_ = *reinterpret_cast<const float4*>(((j) < 4 ? pa : pb) + pipe_pass_off(k, j, B))
    PL(R.r0, 0); PL(R.r1, 1); PL(R.r2, 2); PL(R.r3, 3); PL(R.r4, 4); PL(R.r5, 5); PL(R.r6, 6); PL(R.r7, 7);
#undef PL
}

__device__ __forceinline__ float pipe_sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

// registers -> stage buffer; the row sums of the first region accumulate in rs (the bias gradient)
__device__ __forceinline__ void pipe_regs_store(const PipeRegs& R, float* sd, float (&rs)[4]) {
#define PS(R_, j) *reinterpret_cast<float4*>(sd + 32 * (j) * PIPE_LD) = R_
    PS(R.r0, 0); PS(R.r1, 1); PS(R.r2, 2); PS(R.r3, 3); PS(R.r4, 4); PS(R.r5, 5); PS(R.r6, 6); PS(R.r7, 7);
#undef PS
    rs[0] += pipe_sum4(R.r0); rs[1] += pipe_sum4(R.r1); rs[2] += pipe_sum4(R.r2); rs[3] += pipe_sum4(R.r3);
}

// ---- epilogue waves ----------------------------------------------------------------------------------------------
// where an item's gradient goes
struct PipeDst {
    float* g;         // gradient tensor or null
    NsvdOptPtrs o;    // parameter / square average / EMA shadow (opt != 0)
    float* gb;        // bias gradient tensor or null
    NsvdOptPtrs ob;
};

__device__ __forceinline__ PipeDst pipe_dst(const WgradArgs& a, const PipeItem& p) {
    PipeDst d;
    d.g = a.gW[p.i];
    d.o = a.oW[p.i];
    d.gb = a.gb[p.i];
    d.ob = a.ob[p.i];
    return d;
}

// one float4 group of an item's tile in flight between two slots
struct PipeQuad {
    float4 v, p, sq, ema;
    unsigned off;  // element offset inside the tensor
};

__device__ __forceinline__ int pipe_nquads(const PipeItem& p) { return p.kind == 0 ? 16 : 2; }

template <bool EMA>
__device__ __forceinline__ void pipe_quad_load(PipeQuad& s, const WgradArgs& a, const PipeItem& p, const PipeDst& d,
                                               const float* hand, int q, int ht) {
    if (p.kind == 0) {
        const int row = 8 * q + (ht >> 5), c4 = ht & 31;
        s.v = *reinterpret_cast<const float4*>(hand + row * PIPE_HS_LD + 4 * c4);
        s.off = (unsigned)(((size_t)p.l * HID + row) * (size_t)p.ldw + p.k0 + 4 * c4);
    } else {
        const int row = 32 * q + (ht >> 3), c4 = ht & 7;
        const float4 v0 = *reinterpret_cast<const float4*>(hand + row * PIPE_HS_LD + 4 * c4);
        const float4 v1 = *reinterpret_cast<const float4*>(hand + (64 + row) * PIPE_HS_LD + 4 * c4);
        s.v = make_float4(v0.x + v1.x, v0.y + v1.y, v0.z + v1.z, v0.w + v1.w);
        s.off = (unsigned)(((size_t)p.l * HID + p.n0 + row) * (size_t)p.ldw + p.k0 + 4 * c4);
    }
    if (a.opt) {
        s.p = *reinterpret_cast<const float4*>(d.o.p + s.off);
        s.sq = *reinterpret_cast<const float4*>(d.o.sq + s.off);
        if (EMA) s.ema = *reinterpret_cast<const float4*>(d.o.ema + s.off);
    }
}

template <bool EMA>
__device__ __forceinline__ void pipe_quad_finish(PipeQuad& s, const WgradArgs& a, const PipeDst& d) {
    if (d.g) *reinterpret_cast<float4*>(d.g + s.off) = s.v;
    if (!a.opt) return;
    float e0 = s.ema.x, e1 = s.ema.y, e2 = s.ema.z, e3 = s.ema.w;
    if (!EMA) e0 = e1 = e2 = e3 = 0.f;
    nsvd_rmsprop_upd(s.p.x, s.v.x, s.sq.x, e0, EMA, a.h);
    nsvd_rmsprop_upd(s.p.y, s.v.y, s.sq.y, e1, EMA, a.h);
    nsvd_rmsprop_upd(s.p.z, s.v.z, s.sq.z, e2, EMA, a.h);
    nsvd_rmsprop_upd(s.p.w, s.v.w, s.sq.w, e3, EMA, a.h);
    *reinterpret_cast<float4*>(d.o.p + s.off) = s.p;
    *reinterpret_cast<float4*>(d.o.sq + s.off) = s.sq;
    if (EMA) *reinterpret_cast<float4*>(d.o.ema + s.off) = make_float4(e0, e1, e2, e3);
}

// two float4 groups travel together (a slot of the B-piece shadow is too short a list for one group per slot)
struct PipePair {
    PipeQuad a, b;
};
__device__ __forceinline__ int pipe_npairs(const PipeItem& p) { return pipe_nquads(p) / 2; }

// slot j of the shadowed K loop: finish the pair loaded in the previous slot (one chunk time earlier), load pair j
template <bool EMA>
__device__ __forceinline__ void pipe_slot(PipePair& s, int j, int np, const WgradArgs& a, const PipeItem& p,
                                          const PipeDst& d, const float* hand, int ht) {
    if (j >= 1 && j - 1 < np) {
        pipe_quad_finish<EMA>(s.a, a, d);
        pipe_quad_finish<EMA>(s.b, a, d);
    }
    if (j < np) {
        pipe_quad_load<EMA>(s.a, a, p, d, hand, 2 * j, ht);
        pipe_quad_load<EMA>(s.b, a, p, d, hand, 2 * j + 1, ht);
    }
}

// everything of item p that the slots [0, nslots) did not get to
template <bool EMA>
__device__ __forceinline__ void pipe_drain(PipePair& s, int nslots, const WgradArgs& a, const PipeItem& p,
                                           const PipeDst& d, const float* hand, int ht) {
    const int np = pipe_npairs(p);
    for (int g = nslots - 1 < 0 ? 0 : nslots - 1; g < np; ++g) {
        if (g >= nslots) {
            pipe_quad_load<EMA>(s.a, a, p, d, hand, 2 * g, ht);
            pipe_quad_load<EMA>(s.b, a, p, d, hand, 2 * g + 1, ht);
        }
        pipe_quad_finish<EMA>(s.a, a, d);
        pipe_quad_finish<EMA>(s.b, a, d);
    }
}

// bias gradient of the item whose last chunk has just been staged: rs[j] = this thread's share of the row sums of
// staged rows (t >> 3) + 32 j (the 8 threads t & 7 of a row are lanes of one wave). All state loads are issued
// before the first one is consumed (dependent load-update-store round trips would stall the staging).
__device__ __forceinline__ void pipe_emit_bias(const WgradArgs& a, const PipeItem& p, float (&rs)[4], int ht) {
    if (p.bias) {
#pragma unroll
        for (int off = 1; off < 8; off <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) rs[j] += __shfl_xor(rs[j], off, 64);
        if ((ht & 7) == 0) {
            float* g = a.gb[p.i];
            const NsvdOptPtrs ob = a.ob[p.i];
            const size_t base = (size_t)p.l * HID + p.n0 + (ht >> 3);
            const int n = p.kind == 0 ? 4 : 2;
            float v[4], pv[4], sv[4], ev[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = p.kind == 0 ? rs[j] : rs[j & 1] + rs[(j & 1) + 2];  // kind 1: rows r and 64 + r = batch halves
                pv[j] = sv[j] = ev[j] = 0.f;
                if (a.opt && j < n) {
                    pv[j] = ob.p[base + 32 * j];
                    sv[j] = ob.sq[base + 32 * j];
                    if (ob.ema) ev[j] = ob.ema[base + 32 * j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < n) {
                    if (g) g[base + 32 * j] = v[j];
                    if (a.opt) {
                        nsvd_rmsprop_upd(pv[j], v[j], sv[j], ev[j], ob.ema != nullptr, a.h);
                        ob.p[base + 32 * j] = pv[j];
                        ob.sq[base + 32 * j] = sv[j];
                        if (ob.ema) ob.ema[base + 32 * j] = ev[j];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) rs[j] = 0.f;
}

// the 128 -> 1 layer, db_last and d scales: unit u = (head, 8 rows of the last hidden layer), 2 rows per wave
__device__ __forceinline__ void pipe_last_layer(const WgradArgs& a, int u, int hw, int lane) {
    const int nh = a.nlayers - 1;
    const int l = u >> 4, r0 = 8 * (u & 15) + 2 * hw;
    const float* db = a.dbase + (size_t)l * a.B;
    const float* z0 = a.zsave[nh - 1] + ((size_t)l * HID + r0) * a.B;
    const float* z1 = z0 + a.B;
    const bool head_sums = (u & 15) == 0 && hw == 0;
    float s0 = 0.f, s1 = 0.f, sb = 0.f, ss = 0.f;
    for (int b = 4 * lane; b < a.B; b += 256) {
        const float4 d = *reinterpret_cast<const float4*>(db + b);
        const float4 x0 = *reinterpret_cast<const float4*>(z0 + b);
        const float4 x1 = *reinterpret_cast<const float4*>(z1 + b);
        s0 = fmaf(d.x, x0.x, s0); s0 = fmaf(d.y, x0.y, s0); s0 = fmaf(d.z, x0.z, s0); s0 = fmaf(d.w, x0.w, s0);
        s1 = fmaf(d.x, x1.x, s1); s1 = fmaf(d.y, x1.y, s1); s1 = fmaf(d.z, x1.z, s1); s1 = fmaf(d.w, x1.w, s1);
        if (head_sums) {
            sb += (d.x + d.y) + (d.z + d.w);
            if (a.dfsc) {
                const float4 f = *reinterpret_cast<const float4*>(a.dfsc + (size_t)l * a.B + b);
                ss += (f.x + f.y) + (f.z + f.w);
            }
        }
    }
    s0 = nsvd_wave_sum(s0);
    s1 = nsvd_wave_sum(s1);
    if (head_sums) {
        sb = nsvd_wave_sum(sb);
        ss = nsvd_wave_sum(ss);
    }
    if (lane == 0) {
        const WgDst dW{a.gW[nh], a.opt};
        wg_emit1(a, dW, a.oW[nh], (size_t)l * HID + r0, s0);
        wg_emit1(a, dW, a.oW[nh], (size_t)l * HID + r0 + 1, s1);
        if (head_sums) {
            wg_emit1(a, WgDst{a.gb[nh], a.opt}, a.ob[nh], l, sb);
            if (a.dfsc) wg_emit1(a, WgDst{a.gscales, a.opt}, a.oscales, l, ss);
        }
    }
}

template <bool EMA>
__global__ void __launch_bounds__(PIPE_THREADS, 1) pmlp_wgrad_pipe_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pipe_lds[];
    float* stage = pipe_lds;
    float* hand = pipe_lds + 2 * PIPE_SBUF;
    const int tid = threadIdx.x;
    const bool mfma_wave = tid < 256;
    const int ht = tid - 256;
    const int nItems = pipe_num_items(a);
    const int stride = gridDim.x;
    PIPE_STAMP_M(0);
#ifdef NSVD_WG_STAMPS
    if (threadIdx.x == 0) g_pipe_stamps[(size_t)blockIdx.x * 32 + 31] = wall_clock64();
#endif
    // Barrier protocol (identical counts on both sides, derived from the same item descriptors): per item, one
    // barrier per chunk ("chunk g is staged"), then "free" (the epilogue waves are done with the hand-off tile of the
    // previous item), the MFMA waves write the new tile, then "ready" (it is visible to the epilogue waves).
    if (mfma_wave) {
        int g = 0, k = 0;
        for (int it = blockIdx.x; it < nItems; it += stride, ++k) {
            const PipeItem p = pipe_decode(a, it);
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            if (p.kind == 0) pipe_mfma_item<0>(stage, g, p.nch, acc, tid, k);
            else pipe_mfma_item<1>(stage, g, p.nch, acc, tid, k);
            PIPE_STAMP_M(1 + 3 * (k < 4 ? k : 4));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // free
            if (p.kind == 0) pipe_handoff<0>(acc, hand, tid);
            else pipe_handoff<1>(acc, hand, tid);
            PIPE_STAMP_M(2 + 3 * (k < 4 ? k : 4));
            __syncthreads();  // ready
            PIPE_STAMP_M(3 + 3 * (k < 4 ? k : 4));
        }
        return;
    }
    // ---- epilogue / staging waves
    const size_t B = (size_t)a.B;
    float* sd = stage + (ht >> 3) * PIPE_LD + 4 * (ht & 7);
    PipeCursor ld;    // next chunk to fetch from global memory
    PipeRegs Ra, Rb;  // the chunks one and two ahead of the one the MFMA waves are multiplying
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    pipe_cursor_open(ld, a, blockIdx.x, nItems, ht);
    if (ld.it >= nItems) goto last_layer;  // (grid <= nItems: not reached)
    // prologue: chunk 0 -> LDS buffer 0, chunks 1 and 2 -> registers (Rb, Ra); the cursor stops on the last chunk
#define PIPE_ADVANCE() \
    if (ld.c + 1 < ld.nch || ld.it + stride < nItems) pipe_cursor_next(ld, a, nItems, stride, ht)
    pipe_regs_load(Ra, ld, B);
    PIPE_ADVANCE();
    pipe_regs_load(Rb, ld, B);
    PIPE_ADVANCE();
    pipe_regs_store(Ra, sd, rs);
    pipe_regs_load(Ra, ld, B);
    PIPE_ADVANCE();
    {
        PipeItem p = pipe_decode(a, blockIdx.x), prev;
        prev.kind = -1;
        PipeDst d = pipe_dst(a, p);
        PipePair s0;
        int it = blockIdx.x, c = 0, np = 0, k = 0, g = 0;
        // One step = one chunk barrier and the work behind it. R is the register set holding chunk g + 1: the two
        // call sites below alternate Rb / Ra STATICALLY (a run-time choice between the sets makes the compiler wait
        // for the youngest loads before every store, i.e. lose the two-chunk prefetch distance).
        auto step = [&](PipeRegs& R) -> bool {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk g is staged
            PIPE_STAMP_H(g, 0);
            const bool last = c == p.nch - 1;
            if (last) pipe_emit_bias(a, p, rs, ht);  // the item's row sums are complete
            if (!last || it + stride < nItems) {
                // stage chunk g + 1 (the MFMA waves are done with that buffer)
                pipe_regs_store(R, sd + ((g + 1) & 1) * PIPE_SBUF, rs);
            }
            PIPE_STAMP_H(g, 1);
            pipe_slot<EMA>(s0, c, np, a, prev, d, hand, ht);  // one slot of the previous item's epilogue
            PIPE_STAMP_H(g, 2);
            // fetch chunk g + 3 into the set just stored. Always eight loads, and the youngest ones of the step (past
            // the end of the stream the last chunk is fetched again): the wait in front of the next store of the OTHER
            // set is then exactly vmcnt(8), i.e. the fetch stays two chunk times ahead of its use.
            pipe_regs_load(R, ld, B);
            PIPE_ADVANCE();
            PIPE_STAMP_H(g, 3);
            ++g;
            if (!last) {
                ++c;
                return true;
            }
            PIPE_STAMP_E(16 + 2 * (k < 4 ? k : 4));
            if (prev.kind >= 0) pipe_drain<EMA>(s0, p.nch, a, prev, d, hand, ht);
            PIPE_STAMP_E(17 + 2 * (k < 4 ? k : 4));
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // free: the old tile has been read
            __syncthreads();                                                     // ready: the tile of `it` is in LDS
            prev = p;
            np = pipe_npairs(prev);
            d = pipe_dst(a, prev);
            it += stride;
            ++k;
            c = 0;
            if (it >= nItems) return false;
            p = pipe_decode(a, it);
            return true;
        };
        while (step(Rb) && step(Ra)) {
        }
        PIPE_STAMP_E(26);
        if (prev.kind >= 0) pipe_drain<EMA>(s0, 0, a, prev, d, hand, ht);
    }
#undef PIPE_ADVANCE
last_layer:
    PIPE_STAMP_E(27);
    for (int u = blockIdx.x; u < 16 * a.L; u += gridDim.x) pipe_last_layer(a, u, ht >> 6, ht & 63);
    PIPE_STAMP_E(28);
#ifdef NSVD_WG_STAMPS
    if (threadIdx.x == 256) g_pipe_stamps[(size_t)blockIdx.x * 32 + 30] = wall_clock64();
#endif
}

inline bool pipe_wgrad_ok(const nsvd_model_desc& d, int B, int S) {
    return S == 1 && B % (2 * BK) == 0 && d.nlayers >= 2;
}
