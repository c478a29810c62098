// pmlp_wgrad_pipe_kernel: the weight-gradient kernel as a persistent, software-pipelined workgroup per CU.
// Included by pmlp_bwd.hip (inside its anonymous namespace, after WgradArgs / wg_emit1).
//
// Why: in pmlp_fused_wgrad_kernel every CU runs ONE 128 x 128 dW_0 tile, so all 256 K loops end together and the
// optimiser epilogue (24 B/parameter: read p, sq, ema, write them back) runs with the matrix pipes idle - 10 us of
// a 57 us kernel - while the dW_i quadrants that co-reside with the K loops stretch them from 72 K to 94 K cycles.
// Here a workgroup is 8 waves with three roles, and every CU walks the same short item list - the two 128 x 64 halves
// of its dW_0 tile, then one 64 x 32 piece of a hidden-layer gradient dW_i (contraction split in two halves over the
// wave pairs, so that all 256 CUs share that work evenly):
//   waves 0-3  MFMA waves: fragment reads and MFMAs only; after an item's K loop they drop the accumulators into an
//              LDS hand-off tile and go on with the next item;
//   waves 4-5  staging waves: the global -> register -> LDS copies of every chunk of the item list as ONE continuous
//              stream, two chunks ahead of the multiplications (an item's first chunk is in LDS before the previous
//              item has been handed off). They issue vector LOADS only: on gfx9 loads and stores share vmcnt and
//              complete out of order with respect to each other, so a wave that mixes them can only wait for
//              "everything" - which would pin the prefetch distance to one chunk, below the ~4.5 K-cycle loaded memory
//              latency of this kernel (measured; that is what bounded the two-role versions of this kernel);
//   waves 6-7  epilogue waves: the PREVIOUS item's tile from LDS -> gradient store and / or RMSprop + EMA step while
//              the MFMA waves are in the next K loop (the optimiser traffic of half-tile 1 flies under the K loop of
//              half-tile 2, that of half-tile 2 under the dW_i piece), the bias gradients (row sums handed over by
//              the staging waves) and, at the end, the 128 -> 1 layer, db_last and d scales.
// s_barrier is workgroup-wide on gfx950, so all eight waves execute the same barrier sequence (per item: one barrier
// per 64-row chunk, then "free" and "ready" around the hand-off) and do their role's work between two of them.
// Correctness depends only on the barrier COUNTS agreeing (every role derives them from the same item descriptors),
// never on timing.
// Split-K launches (head-parallel ranks, S > 1) keep the tile kernel above.
#pragma once

constexpr int PIPE_THREADS = 512;
constexpr int PIPE_KC = 64;                            // contraction chunk (rows of the batch per barrier)
constexpr int PIPE_LD = PIPE_KC + 4;                   // padded stage row: conflict-free ds_read_b128 over 16 rows
constexpr int PIPE_HS_LD = 72;                         // hand-off tile row (floats): 64 columns + 8 pad
constexpr int PIPE_SROWS = 192;                        // staged rows per chunk: a 128-row region + a 64-row region
constexpr int PIPE_SBUF = PIPE_SROWS * PIPE_LD;        // one stage buffer (floats)
constexpr int PIPE_HAND = 128 * PIPE_HS_LD;            // the hand-off tile (floats)
constexpr int PIPE_LDS_FLOATS = 2 * PIPE_SBUF + PIPE_HAND + 128;
constexpr size_t PIPE_LDS_BYTES = (size_t)PIPE_LDS_FLOATS * sizeof(float);  // 141 312 B: one workgroup per CU

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
#define PIPE_STAMP_E(slot) if (threadIdx.x == 384) PIPE_STAMP(slot)
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
    int kind;   // 0: half of a dW_0 tile (128 rows x 64 feature columns, K = B);  1: dW_i piece (64 x 32, K = 2 x B/2)
    int l, i;   // head, layer
    int n0, k0; // origin inside W_i[l]: rows n0.., columns k0..
    int ldw;    // row length of W_i[l]
    int nch;    // 64-row chunks each wave contracts over (the K loop has nch + 1 barriers)
    int bias;   // the item also carries the row sums of its dz rows (the bias gradient)
};

__device__ __forceinline__ int pipe_xcd_remap(int idx, int n) {
    // consecutive indices (same head / layer) on the same XCD when the count allows it: blockIdx % 8 is the XCD
    return (n & 7) == 0 ? (idx & 7) * (n >> 3) + (idx >> 3) : idx;
}

__device__ __forceinline__ PipeItem pipe_decode(const WgradArgs& a, int it) {
    PipeItem p;
    const int nA = a.nA;
    if (it < 2 * nA) {
        const int half = it / nA;
        const int unit = pipe_xcd_remap(it - half * nA, nA);
        const int nkt = a.F / HID;
        p.kind = 0;
        p.l = unit / nkt;
        p.i = 0;
        p.n0 = 0;
        p.k0 = (unit - p.l * nkt) * HID + 64 * half;
        p.ldw = a.F;
        p.nch = a.B / PIPE_KC;
        p.bias = p.k0 == 0;
        return p;
    }
    const int nP = 8 * (a.nlayers - 2) * a.L;  // 8 pieces per (layer, head)
    const int q = pipe_xcd_remap(it - 2 * nA, nP);
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
// One chunk = 64 rows of the batch; its stage image is 192 rows of PIPE_LD = 68 floats:
//   kind 0: rows 0..127 = dz_0[l][n][chunk], rows 128..191 = phi^T[k0 + r][chunk]
//   kind 1: rows 0..63 / 64..127 = dz_i[l][n0 + r][chunk of batch half 0 / 1], rows 128..159 / 160..191 =
//           a_{i-1}[l][k0 + r][chunk of batch half 0 / 1]
// Staging thread t (of the 128 of the two staging waves) moves the float4 (row (t >> 4) + 8 j, columns 4 (t & 15)..)
// of pass j = 0..23 (passes 0-15: the 128-row region, 16-23: the 64-row region).
struct PipeSrc {
    const float* pa;   // row (t >> 4) of the 128-row region, this thread's columns of chunk 0
    const float* pb;   // row (t >> 4) of the 64-row region
};

__device__ __forceinline__ PipeSrc pipe_sources(const WgradArgs& a, const PipeItem& p, int ht) {
    PipeSrc s;
    const int s_row = ht >> 4, s_c4 = ht & 15;
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

// element offset of staging pass j relative to pa (j < 16) / pb (j >= 16)
__device__ __forceinline__ size_t pipe_pass_off(int kind, int j, size_t B) {
    if (kind == 0) return (size_t)(8 * (j & 15)) * B;
    if (j < 16) return (size_t)(8 * (j & 7)) * B + (size_t)(j >> 3) * (B / 2);
    return (size_t)(8 * (j & 3)) * B + (size_t)((j - 16) >> 2) * (B / 2);
}

// ---- MFMA waves ------------------------------------------------------------------------------------------------
struct PipeFrag {
    float4 a0, a1, b0;
};

// The nch chunks of one item: fragment reads and MFMAs only (the staging waves keep the two stage buffers filled).
// g = this workgroup's running chunk counter (buffer = g & 1). One barrier per chunk: "chunk g is in LDS"; it sits
// before the LAST q-group of the previous chunk, whose MFMAs cover the first fragment reads of the new one.
template <int KIND>
__device__ __forceinline__ void pipe_mfma_item(const float* stage, int& g, int nch, f32x16 (&acc)[2], int tid, int kk) {
    (void)kk;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    // kind 0: wave (wm, wn) = 64 x 32 of the half tile; kind 1: wave (wm, kh) = 32 x 32 of the piece, batch half kh
    const int arow = KIND == 0 ? 64 * (w >> 1) + li : 64 * (w >> 1) + 32 * (w & 1) + li;
    const int brow = KIND == 0 ? 128 + 32 * (w & 1) + li : 128 + 32 * (w >> 1) + li;
    const float* fa = stage + arow * PIPE_LD + 4 * hi;
    const float* fb = stage + brow * PIPE_LD + 4 * hi;
#define PK_READ(f, buf, q)                                                                             \
    {                                                                                                  \
        f.b0 = *reinterpret_cast<const float4*>(fb + (buf) * PIPE_SBUF + 8 * (q));                     \
        f.a0 = *reinterpret_cast<const float4*>(fa + (buf) * PIPE_SBUF + 8 * (q));                     \
        if (KIND == 0) f.a1 = *reinterpret_cast<const float4*>(fa + (buf) * PIPE_SBUF + 32 * PIPE_LD + 8 * (q)); \
    }
#define PK_MMA1(f, X)                                                                                  \
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.X, f.b0.X, acc[0], 0, 0, 0);                    \
    if (KIND == 0) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.X, f.b0.X, acc[1], 0, 0, 0);
#define PK_MMA(f) PK_MMA1(f, x) PK_MMA1(f, y) PK_MMA1(f, z) PK_MMA1(f, w)
#define PK_FENCE() __builtin_amdgcn_sched_barrier(0)
#define PK_IL(n, mask)                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier((mask), 1, 0);        \
    }
#define PK_NR (KIND == 0 ? 3 : 2)
#define PK_STEP(fr, fm, buf, q) \
    {                           \
        PK_READ(fr, buf, q);    \
        PK_MMA(fm);             \
        PK_IL(PK_NR, 0x100);    \
        PK_FENCE();             \
    }
    PipeFrag f0, f1;
    f0.a1 = f1.a1 = make_float4(0.f, 0.f, 0.f, 0.f);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk g is staged
    PIPE_STAMP_CH(kk, 0);
    PK_READ(f0, g & 1, 0);
    // all chunks but the last: the next chunk's barrier and first fragment reads sit before the last q-group
    for (int c = 0; c + 1 < nch; ++c, ++g) {
        const int cur = g & 1;
        PK_STEP(f1, f0, cur, 1);
        PK_STEP(f0, f1, cur, 2);
        PK_STEP(f1, f0, cur, 3);
        PK_STEP(f0, f1, cur, 4);
        PK_STEP(f1, f0, cur, 5);
        PK_STEP(f0, f1, cur, 6);
        PK_STEP(f1, f0, cur, 7);
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
        PK_STEP(f0, f1, cur, 4);
        PK_STEP(f1, f0, cur, 5);
        PK_STEP(f0, f1, cur, 6);
        PK_STEP(f1, f0, cur, 7);
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
__device__ __forceinline__ void pipe_handoff(const f32x16 (&acc)[2], float* hand, int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    if (KIND == 0) {
        const int wm = w >> 1, wn = w & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hand[(64 * wm + 32 * i + acc_row(r, hi)) * PIPE_HS_LD + 32 * wn + li] = acc[i][r];
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

// The 24 staging registers of one chunk are 24 NAMED local float4 variables per set (an indexed array, or a struct
// passed by reference, ends up in scratch): P##0 .. P##23 for the set prefix P.
#define PIPE_FOR_PASSES(M, P)                                                                                         \
    M(P, 0) M(P, 1) M(P, 2) M(P, 3) M(P, 4) M(P, 5) M(P, 6) M(P, 7) M(P, 8) M(P, 9) M(P, 10) M(P, 11) M(P, 12)        \
    M(P, 13) M(P, 14) M(P, 15) M(P, 16) M(P, 17) M(P, 18) M(P, 19) M(P, 20) M(P, 21) M(P, 22) M(P, 23)
#define PIPE_DECL1(P, j) float4 P##j;
#define PIPE_LOAD1(P, j) \
    P##j = *reinterpret_cast<const float4*>(((j) < 16 ? lpa : lpb) + pipe_pass_off(ld.kind, j, B));
#define PIPE_STORE1(P, j) \
    *reinterpret_cast<float4*>(sdst + ((j) < 16 ? 8 * (j) : 128 + 8 * ((j) - 16)) * PIPE_LD) = P##j;
// fetch the cursor's chunk into set P, then advance the cursor (it stops on the last chunk of the stream, which is then
// fetched again: the load count per step stays fixed, which keeps the compiler's vmcnt bookkeeping exact)
#define PIPE_FETCH(P)                                                                         \
    {                                                                                         \
        const float* lpa = ld.src.pa + (size_t)ld.c * PIPE_KC;                                \
        const float* lpb = ld.src.pb + (size_t)ld.c * PIPE_KC;                                \
        PIPE_FOR_PASSES(PIPE_LOAD1, P)                                                        \
        if (ld.c + 1 < ld.nch || ld.it + stride < nItems) pipe_cursor_next(ld, a, nItems, stride, st); \
    }
#define PIPE_PUT(P, buf)                                  \
    {                                                     \
        float* sdst = sd + (buf) * PIPE_SBUF;             \
        PIPE_FOR_PASSES(PIPE_STORE1, P)                   \
    }

__device__ __forceinline__ float pipe_sum4(const float4& v) { return (v.x + v.y) + (v.z + v.w); }

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

// float4 groups per epilogue thread (128 of them): kind 0: 128 x 64 / 4 / 128 = 16; kind 1: 64 x 32 / 4 / 128 = 4
__device__ __forceinline__ int pipe_nquads(const PipeItem& p) { return p.kind == 0 ? 16 : 4; }

template <bool EMA>
__device__ __forceinline__ void pipe_quad_load(PipeQuad& s, const WgradArgs& a, const PipeItem& p, const PipeDst& d,
                                               const float* hand, int q, int ht) {
    if (p.kind == 0) {
        const int row = 8 * q + (ht >> 4), c4 = ht & 15;
        s.v = *reinterpret_cast<const float4*>(hand + row * PIPE_HS_LD + 4 * c4);
        s.off = (unsigned)(((size_t)p.l * HID + row) * (size_t)p.ldw + p.k0 + 4 * c4);
    } else {
        const int row = 16 * q + (ht >> 3), c4 = ht & 7;
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

// Bias gradient = row sums of the item's dz rows (the 128-row stage region). The epilogue waves add them up from the
// stage buffer while the chunk is there (thread et owns stage row et; only items that carry a bias do this), park them
// in hb[] before the item's "free" barrier and emit them with the item's epilogue.
__device__ __forceinline__ float pipe_bias_chunk(const float* stage_buf, int et) {
    const float* r = stage_buf + et * PIPE_LD;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < PIPE_KC / 8; ++q) {
        const float4 u = *reinterpret_cast<const float4*>(r + 8 * q);
        const float4 v = *reinterpret_cast<const float4*>(r + 8 * q + 4);
        s0 += (u.x + u.y) + (u.z + u.w);
        s1 += (v.x + v.y) + (v.z + v.w);
    }
    return s0 + s1;
}

// epilogue side: thread et < 128 owns row et of the item's dz rows
__device__ __forceinline__ void pipe_emit_bias(const WgradArgs& a, const PipeItem& p, const float* hb, int et) {
    if (!p.bias) return;
    const WgDst db{a.gb[p.i], a.opt};
    if (p.kind == 0) {
        wg_emit1(a, db, a.ob[p.i], (size_t)p.l * HID + et, hb[et]);
    } else if (et < 64) {  // rows r and 64 + r are the two batch halves of row n0 + r
        wg_emit1(a, db, a.ob[p.i], (size_t)p.l * HID + p.n0 + et, hb[et] + hb[64 + et]);
    }
}

// the 128 -> 1 layer, db_last and d scales: unit u = (head, 4 rows of the last hidden layer), 2 rows per epilogue wave
__device__ __forceinline__ void pipe_last_layer(const WgradArgs& a, int u, int hw, int lane) {
    const int nh = a.nlayers - 1;
    const int l = u >> 5, r0 = 4 * (u & 31) + 2 * hw;
    const float* db = a.dbase + (size_t)l * a.B;
    const float* z0 = a.zsave[nh - 1] + ((size_t)l * HID + r0) * a.B;
    const float* z1 = z0 + a.B;
    const bool head_sums = (u & 31) == 0 && hw == 0;
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
    const int nItems = 2 * a.nA + 8 * (a.nlayers - 2) * a.L;
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
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
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
    float* hb = hand + PIPE_HAND;  // [128] row sums of the item being handed over
    if (tid < 384) {
        // ---- staging waves (4, 5): vector loads only
        const int st = tid - 256;
        const size_t B = (size_t)a.B;
        float* sd = stage + (st >> 4) * PIPE_LD + 4 * (st & 15);
        PipeCursor ld;    // next chunk to fetch from global memory
        PIPE_FOR_PASSES(PIPE_DECL1, ra)  // the chunks one and two ahead of the one the MFMA waves are multiplying
        PIPE_FOR_PASSES(PIPE_DECL1, rb)
        pipe_cursor_open(ld, a, blockIdx.x, nItems, st);
        // prologue: chunk 0 -> LDS buffer 0, chunks 1 and 2 -> registers (rb, then ra: the same order of outstanding
        // loads as at the loop's back edge, so that the compiler's vmcnt bookkeeping stays exact inside the loop)
        PIPE_FETCH(ra);
        PIPE_PUT(ra, 0);
        PIPE_FETCH(rb);
        PIPE_FETCH(ra);
        PipeItem p = pipe_decode(a, blockIdx.x);
        int it = blockIdx.x, c = 0, g = 0;
        bool more = true;
        // One step = one chunk barrier and the staging behind it: store the set holding chunk g + 1 (the MFMA waves are
        // done with that buffer), fetch chunk g + 3 into it. The two sets alternate STATICALLY (token pasting).
#define PIPE_STAGE_STEP(P)                                                                   \
    {                                                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* chunk g staged */ \
        PIPE_STAMP_H(g, 0);                                                                  \
        const bool last = c == p.nch - 1;                                                    \
        if (!last || it + stride < nItems) PIPE_PUT(P, (g + 1) & 1);                         \
        PIPE_STAMP_H(g, 1);                                                                  \
        PIPE_FETCH(P);                                                                       \
        PIPE_STAMP_H(g, 2);                                                                  \
        ++g;                                                                                 \
        if (!last) {                                                                         \
            ++c;                                                                             \
        } else {                                                                             \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* free */       \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); /* ready */      \
            it += stride;                                                                    \
            c = 0;                                                                           \
            more = it < nItems;                                                              \
            if (more) p = pipe_decode(a, it);                                                \
        }                                                                                    \
    }
        // (unrolled four deep: the compiler resets its count of outstanding loads to "all of them" at the loop header,
        //  so only the first step of the body waits for more than it needs)
        while (more) {
            PIPE_STAGE_STEP(rb);
            if (!more) break;
            PIPE_STAGE_STEP(ra);
            if (!more) break;
            PIPE_STAGE_STEP(rb);
            if (!more) break;
            PIPE_STAGE_STEP(ra);
        }
#undef PIPE_STAGE_STEP
        return;
    }
    // ---- epilogue waves (6, 7)
    {
        const int et = tid - 384;
        PipeItem p = pipe_decode(a, blockIdx.x), prev;
        prev.kind = -1;
        PipeDst d = pipe_dst(a, p);
        PipePair s0;
        int np = 0, k = 0, g = 0;
        for (int it = blockIdx.x; it < nItems; it += stride, ++k) {
            p = pipe_decode(a, it);
            float bsum = 0.f;
            for (int c = 0; c < p.nch; ++c, ++g) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // chunk g is staged
                if (c == 0 && prev.kind >= 0) pipe_emit_bias(a, prev, hb, et);
                if (p.bias) bsum += pipe_bias_chunk(stage + (g & 1) * PIPE_SBUF, et);
                pipe_slot<EMA>(s0, c, np, a, prev, d, hand, et);  // one slot of the previous item's epilogue
            }
            PIPE_STAMP_E(16 + 2 * (k < 4 ? k : 4));
            if (prev.kind >= 0) pipe_drain<EMA>(s0, p.nch, a, prev, d, hand, et);
            PIPE_STAMP_E(17 + 2 * (k < 4 ? k : 4));
            hb[et] = bsum;  // (the previous item's sums were consumed at this item's first chunk)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // free: the old tile has been read
            __syncthreads();                                                     // ready: the tile of `it` is in LDS
            prev = p;
            np = pipe_npairs(prev);
            d = pipe_dst(a, prev);
        }
        PIPE_STAMP_E(26);
        if (prev.kind >= 0) {
            pipe_emit_bias(a, prev, hb, et);
            pipe_drain<EMA>(s0, 0, a, prev, d, hand, et);
        }
        PIPE_STAMP_E(27);
        for (int u = blockIdx.x; u < 32 * a.L; u += gridDim.x) pipe_last_layer(a, u, et >> 6, et & 63);
        PIPE_STAMP_E(28);
#ifdef NSVD_WG_STAMPS
        if (threadIdx.x == 384) g_pipe_stamps[(size_t)blockIdx.x * 32 + 30] = wall_clock64();
#endif
    }
}

inline bool pipe_wgrad_ok(const nsvd_model_desc& d, int B, int S) {
    return S == 1 && B % (2 * PIPE_KC) == 0 && d.nlayers >= 2;
}
