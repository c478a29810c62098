// pmlp_wgrad_pipe_kernel: the weight-gradient kernel as a persistent, software-pipelined workgroup per CU.
// Included by pmlp_bwd.hip (inside its anonymous namespace, after WgradArgs / wg_emit1).
//
// Why: in pmlp_fused_wgrad_kernel every CU runs ONE 128 x 128 dW_0 tile, so all 256 K loops end together and the
// optimiser epilogue (24 B/parameter: read p, sq, ema, write them back) runs with the matrix pipes idle - 10 us of
// a 57 us kernel - while the dW_i quadrants that co-reside with the K loops stretch them from 72 K to 94 K cycles.
// Here a workgroup is 8 waves:
//   waves 0-3  "MFMA waves": walk a static list of items - the two 128 x 64 halves of a dW_0 tile, then one
//              64 x 32 piece of a hidden-layer gradient dW_i (contraction split in two halves over the wave pairs, so
//              that all 256 CUs share that work evenly) - and after each K loop drop the accumulators into an LDS
//              hand-off tile;
//   waves 4-7  "epilogue waves": take the PREVIOUS item's tile from LDS and store the gradient and / or take the
//              RMSprop + EMA step on its parameters while the MFMA waves are in the next K loop: the optimiser
//              traffic of half-tile 1 flies under the K loop of half-tile 2, that of half-tile 2 under the dW_i piece.
// s_barrier is workgroup-wide on gfx950, so the epilogue waves execute exactly the barriers of the K loop they
// shadow (one per 32-row chunk) and do one slot of their work between two of them: state loads are issued two
// slots (two chunk times, ~4 K cycles) before they are consumed. Correctness depends only on the barrier COUNTS
// agreeing (both sides derive them from the same item descriptor), never on timing.
// The 128 -> 1 layer, db_last and d scales (FMA reductions over the batch) are done by the epilogue waves at the end.
// Split-K launches (head-parallel ranks, S > 1) keep the tile kernel above.
#pragma once

constexpr int PIPE_THREADS = 512;
constexpr int PIPE_HS_LD = 72;                         // hand-off tile row (floats): 64 columns + 8 pad
constexpr int PIPE_SROWS = 192;                        // staged rows per chunk: a 128-row region + a 64-row region
constexpr int PIPE_SBUF = PIPE_SROWS * A_LD;           // one stage buffer (floats)
constexpr int PIPE_HAND = 128 * PIPE_HS_LD;            // one hand-off tile (floats)
constexpr int PIPE_LDS_FLOATS = 2 * PIPE_SBUF + 2 * PIPE_HAND + 2 * 128;
constexpr size_t PIPE_LDS_BYTES = (size_t)PIPE_LDS_FLOATS * sizeof(float);  // 130 048 B: one workgroup per CU

struct PipeItem {
    int kind;   // 0: half of a dW_0 tile (128 rows x 64 feature columns, K = B);  1: dW_i piece (64 x 32, K = 2 x B/2)
    int l, i;   // head, layer
    int n0, k0; // origin inside W_i[l]: rows n0.., columns k0..
    int ldw;    // row length of W_i[l]
    int nch;    // 32-row chunks each wave contracts over (the K loop has nch + 1 barriers)
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
        p.nch = a.B / BK;
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
    p.nch = a.B / (2 * BK);
    p.bias = p.k0 == 0;
    return p;
}

// ---- MFMA waves ------------------------------------------------------------------------------------------------
// Stage layout of one chunk (rows of A_LD = 36 floats, 32 contraction columns each):
//   kind 0: rows 0..127 = dz_0[l][n][chunk], rows 128..191 = phi^T[k0 + r][chunk]
//   kind 1: rows 0..63 / 64..127 = dz_i[l][n0 + r][chunk of batch half 0 / 1], rows 128..159 / 160..191 = a_{i-1}[l][k0 + r][..]
// so that the staging code (6 float4 per thread and chunk) is the same for both kinds.
struct PipeSrc {  // named members (arrays of pointers end up in scratch)
    const float *a0, *a1, *a2, *a3, *b0, *b1;
};

__device__ __forceinline__ PipeSrc pipe_sources(const WgradArgs& a, const PipeItem& p, int s_row, int s_c4) {
    PipeSrc s;
    const size_t B = (size_t)a.B;
    if (p.kind == 0) {
        s.a0 = a.dz[0] + ((size_t)p.l * HID + s_row) * B + 4 * s_c4;
        s.a1 = s.a0 + 32 * B;
        s.a2 = s.a0 + 64 * B;
        s.a3 = s.a0 + 96 * B;
        s.b0 = a.phiTc + ((size_t)p.k0 + s_row) * B + 4 * s_c4;
        s.b1 = s.b0 + 32 * B;
    } else {
        s.a0 = a.dz[p.i] + ((size_t)p.l * HID + p.n0 + s_row) * B + 4 * s_c4;
        s.a1 = s.a0 + 32 * B;
        s.a2 = s.a0 + B / 2;
        s.a3 = s.a1 + B / 2;
        s.b0 = a.zsave[p.i - 1] + ((size_t)p.l * HID + p.k0 + s_row) * B + 4 * s_c4;
        s.b1 = s.b0 + B / 2;
    }
    return s;
}

template <int KIND>
__device__ __forceinline__ void pipe_kloop(const PipeSrc& src, int nch, float* stage, f32x16 (&acc)[2], float (&rs)[4],
                                           int tid) {
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int s_row = tid >> 3, s_c4 = tid & 7;
    // fragment rows of this wave: kind 0: wave (wm, wn) = 64 x 32 of the half tile; kind 1: wave (wm, kh) = 32 x 32
    // of the piece over batch half kh
    const int arow = KIND == 0 ? 64 * (w >> 1) + li : 64 * (w >> 1) + 32 * (w & 1) + li;
    const int brow = KIND == 0 ? 128 + 32 * (w & 1) + li : 128 + 32 * (w >> 1) + li;
    float4 ra0, ra1, ra2, ra3, rb0, rb1;  // named: an indexed array of staging registers ends up in scratch
#define PIPE_LOAD(c)                                                          \
    {                                                                         \
        ra0 = *reinterpret_cast<const float4*>(src.a0 + (size_t)(c) * BK);    \
        ra1 = *reinterpret_cast<const float4*>(src.a1 + (size_t)(c) * BK);    \
        ra2 = *reinterpret_cast<const float4*>(src.a2 + (size_t)(c) * BK);    \
        ra3 = *reinterpret_cast<const float4*>(src.a3 + (size_t)(c) * BK);    \
        rb0 = *reinterpret_cast<const float4*>(src.b0 + (size_t)(c) * BK);    \
        rb1 = *reinterpret_cast<const float4*>(src.b1 + (size_t)(c) * BK);    \
    }
#define PIPE_STORE(buf)                                                       \
    {                                                                         \
        float* d_ = stage + (buf) * PIPE_SBUF + s_row * A_LD + 4 * s_c4;      \
        *reinterpret_cast<float4*>(d_) = ra0;                                 \
        *reinterpret_cast<float4*>(d_ + 32 * A_LD) = ra1;                     \
        *reinterpret_cast<float4*>(d_ + 64 * A_LD) = ra2;                     \
        *reinterpret_cast<float4*>(d_ + 96 * A_LD) = ra3;                     \
        *reinterpret_cast<float4*>(d_ + 128 * A_LD) = rb0;                    \
        *reinterpret_cast<float4*>(d_ + 160 * A_LD) = rb1;                    \
        rs[0] += (ra0.x + ra0.y) + (ra0.z + ra0.w);                           \
        rs[1] += (ra1.x + ra1.y) + (ra1.z + ra1.w);                           \
        rs[2] += (ra2.x + ra2.y) + (ra2.z + ra2.w);                           \
        rs[3] += (ra3.x + ra3.y) + (ra3.z + ra3.w);                           \
    }
    PIPE_LOAD(0);
    PIPE_STORE(0);
    __syncthreads();  // barrier 0 of nch + 1
    if (nch > 1) PIPE_LOAD(1);
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1;
        const float* Ap = stage + cur * PIPE_SBUF + arow * A_LD + 4 * hi;
        const float* Bp = stage + cur * PIPE_SBUF + brow * A_LD + 4 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 b0 = *reinterpret_cast<const float4*>(Bp + 8 * q);
            const float4 a0 = *reinterpret_cast<const float4*>(Ap + 8 * q);
            if (KIND == 0) {
                const float4 a1 = *reinterpret_cast<const float4*>(Ap + 32 * A_LD + 8 * q);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0.x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b0.y, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b0.z, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b0.w, acc[1], 0, 0, 0);
            } else {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc[0], 0, 0, 0);
            }
            if (q == 1 && c + 1 < nch) PIPE_STORE(cur ^ 1);  // chunk c + 1 into the other buffer, mid-chunk
        }
        __syncthreads();  // barrier c + 1
        if (c + 2 < nch) PIPE_LOAD(c + 2);
    }
#undef PIPE_LOAD
#undef PIPE_STORE
}

// accumulators -> hand-off tile, row sums -> hb[]
template <int KIND>
__device__ __forceinline__ void pipe_handoff(const f32x16 (&acc)[2], float (&rs)[4], float* hand, float* hb, int tid,
                                             bool bias) {
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
    if (bias) {
        // 8 threads (s_c4) hold partial sums of staged rows s_row + 32 j
#pragma unroll
        for (int off = 1; off < 8; off <<= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) rs[j] += __shfl_xor(rs[j], off, 64);
        if ((tid & 7) == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) hb[(tid >> 3) + 32 * j] = rs[j];
        }
    }
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

__device__ __forceinline__ int pipe_nquads(const PipeItem& p) { return p.kind == 0 ? 8 : 2; }

template <bool EMA>
__device__ __forceinline__ void pipe_quad_load(PipeQuad& s, const WgradArgs& a, const PipeItem& p, const PipeDst& d,
                                               const float* hand, int q, int ht) {
    if (p.kind == 0) {
        const int row = 16 * q + (ht >> 4), c4 = ht & 15;
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

// slot j of the shadowed K loop: finish the group loaded two slots ago, load group j
template <bool EMA>
__device__ __forceinline__ void pipe_slot(PipeQuad& s, int j, int nq, const WgradArgs& a, const PipeItem& p,
                                          const PipeDst& d, const float* hand, int ht) {
    if (j >= 2 && j - 2 < nq) pipe_quad_finish<EMA>(s, a, d);
    if (j < nq) pipe_quad_load<EMA>(s, a, p, d, hand, j, ht);
}

// everything of item p that the slots [0, nslots) did not get to, and its bias gradient
template <bool EMA>
__device__ __forceinline__ void pipe_drain(PipeQuad& s0, PipeQuad& s1, int nslots, const WgradArgs& a,
                                           const PipeItem& p, const PipeDst& d, const float* hand, const float* hb,
                                           int ht) {
    const int nq = pipe_nquads(p);
    int q = nslots - 2 < 0 ? 0 : nslots - 2;
    for (; q < nq; ++q) {
        // (no reference selected at run time: that would put both groups in scratch)
        if (q & 1) {
            if (q >= nslots) pipe_quad_load<EMA>(s1, a, p, d, hand, q, ht);
            pipe_quad_finish<EMA>(s1, a, d);
        } else {
            if (q >= nslots) pipe_quad_load<EMA>(s0, a, p, d, hand, q, ht);
            pipe_quad_finish<EMA>(s0, a, d);
        }
    }
    if (p.bias) {
        const WgDst db{d.gb, a.opt};
        if (p.kind == 0) {
            if (ht < HID) wg_emit1(a, db, d.ob, (size_t)p.l * HID + ht, hb[ht]);
        } else {
            if (ht < 64) wg_emit1(a, db, d.ob, (size_t)p.l * HID + p.n0 + ht, hb[ht] + hb[64 + ht]);
        }
    }
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
    float* hand0 = pipe_lds + 2 * PIPE_SBUF;
    float* hb0 = hand0 + 2 * PIPE_HAND;
    const int tid = threadIdx.x;
    const bool mfma_wave = tid < 256;
    const int ht = tid - 256;
    const int nItems = 2 * a.nA + 8 * (a.nlayers - 2) * a.L;
    PipeItem prev;
    prev.kind = -1;
    PipeQuad s0, s1;
    int k = 0;  // items done by this workgroup
    for (int it = blockIdx.x; it < nItems; it += gridDim.x, ++k) {
        const PipeItem p = pipe_decode(a, it);
        float* hand = hand0 + (k & 1) * PIPE_HAND;
        float* hb = hb0 + (k & 1) * 128;
        if (mfma_wave) {
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            float rs[4] = {0.f, 0.f, 0.f, 0.f};
            const PipeSrc src = pipe_sources(a, p, tid >> 3, tid & 7);
            if (p.kind == 0) {
                pipe_kloop<0>(src, p.nch, stage, acc, rs, tid);
                pipe_handoff<0>(acc, rs, hand, hb, tid, p.bias != 0);
            } else {
                pipe_kloop<1>(src, p.nch, stage, acc, rs, tid);
                pipe_handoff<1>(acc, rs, hand, hb, tid, p.bias != 0);
            }
        } else {
            // shadow the K loop of item `it` (p.nch + 1 barriers) with the epilogue of the previous item
            const int nslots = p.nch + 1;
            if (prev.kind >= 0) {
                const float* ph = hand0 + ((k - 1) & 1) * PIPE_HAND;
                const float* phb = hb0 + ((k - 1) & 1) * 128;
                const PipeDst d = pipe_dst(a, prev);
                const int nq = pipe_nquads(prev);
                for (int j = 0; j < nslots; j += 2) {
                    pipe_slot<EMA>(s0, j, nq, a, prev, d, ph, ht);
                    __builtin_amdgcn_s_barrier();
                    if (j + 1 < nslots) {
                        pipe_slot<EMA>(s1, j + 1, nq, a, prev, d, ph, ht);
                        __builtin_amdgcn_s_barrier();
                    }
                }
                pipe_drain<EMA>(s0, s1, nslots, a, prev, d, ph, phb, ht);
            } else {
                for (int j = 0; j < nslots; ++j) __builtin_amdgcn_s_barrier();
            }
        }
        __syncthreads();  // hand-off: the tile of item `it` is in LDS
        prev = p;
    }
    if (mfma_wave) return;
    if (prev.kind >= 0) {
        const PipeDst d = pipe_dst(a, prev);
        pipe_drain<EMA>(s0, s1, 0, a, prev, d, hand0 + ((k - 1) & 1) * PIPE_HAND, hb0 + ((k - 1) & 1) * 128, ht);
    }
    for (int u = blockIdx.x; u < 16 * a.L; u += gridDim.x) pipe_last_layer(a, u, ht >> 6, ht & 63);
}

inline bool pipe_wgrad_ok(const nsvd_model_desc& d, int B, int S) {
    return S == 1 && B % (2 * BK) == 0 && d.nlayers >= 2;
}
