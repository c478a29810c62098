// Shared by the two translation units of the fused MFMA path (pmlp_fwd.hip, pmlp_bwd.hip): tile constants, the
// accumulator layout of v_mfma_f32_32x32x2_f32, and the workspace layout.
#pragma once
#include <string.h>
#include "nsvd_kernels.h"

namespace nsvd_pmlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HID = 128;      // hidden width
constexpr int BS = 32;        // base samples per workgroup
constexpr int BK = 32;        // layer-0 K chunk
constexpr int A_LD = BK + 4;  // padded row of the W_0 tile (floats)

// accumulator register r of lane-half hi holds row (r&3) + 8 (r>>2) + 4 hi of the 32-row tile
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// One q-group = 8 consecutive k: fragment loads (one ds_read_b128 per 32-row tile: 4 k's for each of the
// two lane halves) and the 4 x E MFMAs that consume them.
template <int E>
struct Frag {
    float4 a;
    float4 b[E];
};

template <int E>
__device__ __forceinline__ void load_frag(Frag<E>& f, const float* Ap, const float* Bp, int ldb) {
    f.a = *reinterpret_cast<const float4*>(Ap);
#pragma unroll
    for (int e = 0; e < E; ++e) f.b[e] = *reinterpret_cast<const float4*>(Bp + e * BS * ldb);
}

template <int E>
__device__ __forceinline__ void mma_frag(f32x16 (&acc)[E], const Frag<E>& f) {
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.x, f.b[e].x, acc[e], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.y, f.b[e].y, acc[e], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.z, f.b[e].z, acc[e], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < E; ++e) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a.w, f.b[e].w, acc[e], 0, 0, 0);
}

constexpr int H_LD = HID + 4;  // padded row of the activation image [column][k]

// Workgroup -> (head, sample block) of the forward kernel and of the backward chain kernel. Blocks b and b + 8 share an
// XCD (round-robin dispatch: speed only, never correctness). XCD x gets the head group x % HX and the sample-block
// group x / HX, so its L2 sees L / HX weight slabs and nsb / SX feature slabs instead of everything (HX * SX = 8); the
// chain kernel uses the SAME map, so that the activations a forward workgroup saved are read back through the L2
// they were written through and a head's hidden-layer weights cross the fabric once per XCD group, not eight times.
// HX = 0: plain mapping (head-major).
inline int pick_xcd_remap(int L, int nsb, int F) {
    double best = 1e300;
    int pick = 0;
    for (int HX = 1; HX <= 8; HX *= 2) {
        const int SX = 8 / HX;
        if (L % HX != 0 || nsb % SX != 0) continue;
        // (the feature slab is the centre rows only: the stencil rows are generated in the kernel)
        const double bytes = (double)(L / HX) * HID * F + (double)(nsb / SX) * BS * F;
        if (bytes < best) {
            best = bytes;
            pick = HX;
        }
    }
    return pick;
}
// the same choice for the dW_0 tiles of the weight-gradient kernel: XCD = (head group) x (feature-tile group); a
// tile reads dz_0 of its head (128 rows of the batch) and its tw rows of phi^T, and the 32 tiles of an XCD should
// share as many of those rows as their L2 can keep (cfg2: 4 heads x 8 feature tiles = 3 MB instead of 2 x 16 = 4.5 MB)
inline int pick_xcd_remap_wgrad(int L, int nkt, int tw) {
    double best = 1e300;
    int pick = 0;
    for (int HX = 1; HX <= 8; HX *= 2) {
        const int KX = 8 / HX;
        if (L % HX != 0 || nkt % KX != 0) continue;
        const double rows = (double)(L / HX) * HID + (double)(nkt / KX) * tw;
        if (rows < best) {
            best = rows;
            pick = HX;
        }
    }
    return pick;
}
__device__ __forceinline__ void xcd_block_map(int block, int HX, int L, int nsb, int& l, int& sb) {
    if (HX) {
        const int SX = 8 / HX;
        const int x = block & 7, slot = block >> 3;
        const int hpg = L / HX, spg = nsb / SX;  // heads / sample blocks per group
        l = (x % HX) * hpg + slot % hpg;
        sb = (x / HX) * spg + slot / hpg;
    } else {
        l = block / nsb;
        sb = block - l * nsb;
    }
}

struct FusedWs {
    float* phi;                       // (B, F) sample-major Fourier features of the centre rows
    float* sctab;                     // (D, 2, m) cos / sin of eps * fourier_B + (D, m) cos - 1 (even / odd stencil rows)
    float* phiTc;                     // (F, B) feature-major copy of the centre rows (weight gradient)
    float* zsave[NSVD_MAX_LAYERS];    // (L, 128, B) per hidden layer
    float* jac;                       // (B, L)
    float* dsc;                       // (B, L)
    float* dz[NSVD_MAX_LAYERS];       // (L, 128, B) per hidden layer
    float* dbase;                     // (L, B)
    float* dfsc;                      // (L, B)
    float* gpart;                     // (S, slice) split-K partial gradients, S = wgrad_slices() > 1 only
    unsigned short* w0p;              // (3, L, 128, F) bf16 planes of W_0, fragment-major (NSVD_PATH_FUSED_BF16X3 only)
    unsigned short* whp;              // (nlayers - 2, 3, L, 128, 128) bf16 planes of W_1 .. (the same path)
    float* base_raw;                  // (L, (1 + 2D) B) head outputs per stencil point (split-stencil forward only)
    float* loss_part;                 // (L, 32 + 1) partial sums of the loss (direct-moment backward, B <= 1024)
    float* kpart;                     // K-split forward only (fwd_kslices() > 1): the slices' partial layer-0 pre-activations
    unsigned* tickets;                // K-split forward only: one arrival counter per (head, 32-sample block) of the second
                                      // launch (zeroed by the first): the direction that arrives last forms f, Tf
    size_t bytes;
};

// Shape of the layer-0 weight-gradient work: dW_0 tiles of 128 hidden units x `tw` features, the batch contraction cut
// into S slices (split-K with a second pass, wgrad_reduce_kernel, that adds the slices in order and applies the
// optimiser). Measured on MI355X, us per training step (oscillator model, (L, m, B); nA = 128-wide tile count):
//   (S, tw) =            (1,128) (1,64) (2,128) (2,64) (4,128) (4,64)
//   (32, 256, 1024) nA 128  462    443    469     473    483     483
//   (16, 256,  512) nA  64  134    119    139     134    139     143
//   (16, 256, 4096) nA  64  880    782    822     778    780     784
//   ( 4, 1024, 1024) nA 64  280    249    258     253    251     257
//   ( 8, 256, 1024) nA  32  163    133    141     130    137     129
//   ( 2, 1024, 4096) nA 32  465    351    354     299    301     293
// (cfg2 itself, nA 256: 64-wide tiles = 512 of them, two per CU and the small workgroups in a second round: 57.1 us
// for the kernel against 52.7.)
// Hence: 64-wide tiles whenever that still leaves at most 256 of them (twice the tiles at no cost: no second pass);
// the batch is doubled into slices only while fewer than half the CUs have a tile, and beyond that only while a slice
// still keeps 1024 rows (the second pass moves the whole parameter set once more); a slice never drops under 256
// rows (8 chunks, to amortise a tile's prologue and epilogue).
inline int wgrad_tile_width(int nA128, int S) { return nA128 * S <= 128 ? 64 : 128; }
inline int wgrad_slices(const nsvd_model_desc& d, int B) {
    const int nA = (2 * d.m / HID) * d.L;
    int S = 1;
    for (;;) {
        const int rows = B / (2 * S);  // rows per slice after one more doubling
        if (S >= 16 || rows % BK != 0 || rows < 256) break;
        const int tiles = nA * S * (128 / wgrad_tile_width(nA, S));
        if (!(tiles < 128 || (tiles < 256 && rows >= 1024))) break;
        S *= 2;
    }
    return S;
}

// The streaming backward (pmlp_stream_bwd.h: one workgroup = head x batch slice, no dz through HBM): plain-model shapes
// with F = 128 features and two hidden layers, enough (head, 32-sample chunk) pairs to keep a slice long. Returns the
// number of batch slices (heads x slices >= 256 workgroups where the batch allows), 0 = the two-launch form.
inline int stream_bwd_slices(const nsvd_model_desc& d, int B) {
    if (d.nlayers != 3 || 2 * d.m != HID || d.dims[0] != HID || d.dims[1] != HID) return 0;
    if (B % BS != 0 || (B / BS) * d.L < 2048) return 0;
    int S = 1;
    while (d.L * S < 256 && S < 16) S *= 2;
    while (S > 1 && (B % (BS * S) != 0 || B / S < 256)) S /= 2;
    if (B / S < 256) return 0;
    return S;
}

// per-slice layout of the partial gradients: [W_0 | .. | W_n | b_0 | .. | b_n | scales], each padded to 4 floats
struct PartLayout {
    size_t oW[NSVD_MAX_LAYERS], ob[NSVD_MAX_LAYERS], oscales, nW[NSVD_MAX_LAYERS], nb[NSVD_MAX_LAYERS], nscales;
    size_t stride;
};
inline PartLayout part_layout(const nsvd_model_desc& d) {
    PartLayout p;
    memset(&p, 0, sizeof(p));
    size_t off = 0;
    auto put = [&](size_t n) { const size_t o = off; off += (n + 3) / 4 * 4; return o; };
    for (int i = 0; i < d.nlayers; ++i) {
        p.nW[i] = (size_t)d.L * d.dims[i] * (i == 0 ? 2 * (size_t)d.m : (size_t)d.dims[i - 1]);
        p.oW[i] = put(p.nW[i]);
    }
    for (int i = 0; i < d.nlayers; ++i) {
        p.nb[i] = (size_t)d.L * d.dims[i];
        p.ob[i] = put(p.nb[i]);
    }
    p.nscales = d.has_exp_mask ? (size_t)d.L : 0;
    p.oscales = put(p.nscales);
    p.stride = (off + 63) / 64 * 64;
    return p;
}

// K-split of the forward's layer 0 (pmlp_fwd.hip, template parameter KS): a D = 2 stencil batch that leaves most CUs
// idle (configs[0]: 64 workgroups of the plain form) cuts the layer-0 contraction - nine tenths of the kernel - into K
// slices of one workgroup each, which leave their partial pre-activations in the workspace; a second launch adds them in
// slice order and runs the rest of the network. Returns the slice count: 4 = plain form (all five stencil tiles per
// workgroup, 4 x 64 = 256 workgroups, no fd_epilogue launch), 2 = on top of the split-stencil form (one direction per
// workgroup, 2 x 2 x 64: narrower feature blocks), 1 = no K-split.
inline int fwd_kslices(const nsvd_model_desc& d, int B) {
    if (d.D != 2 || B % BS != 0) return 1;
    if ((B / BS) * d.L > 64) return 1;
    if (d.m % 128 == 0 && d.m >= 256) return 4;  // a slice is an even number (>= 4) of 32-wide chunks
    if (d.m % 64 == 0 && d.m >= 128) return 2;
    return 1;
}
inline size_t fwd_kpart_floats(const nsvd_model_desc& d, int B) {
    const int ks = fwd_kslices(d, B);
    if (ks == 4) return (size_t)4 * (B / BS) * d.L * 5 * HID * BS;      // slices x workgroups x (5 tiles of 128 x 32)
    if (ks == 2) return (size_t)2 * 2 * (B / BS) * d.L * 3 * HID * BS;  // slices x (2 directions x workgroups) x 3 tiles
    return 0;
}

inline FusedWs carve_fused(const nsvd_model_desc& d, int B, void* base) {
    FusedWs w;
    memset(&w, 0, sizeof(w));
    const size_t F = 2 * (size_t)d.m;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float* q = (float*)(p + off);
        off += nsvd_align(nfloats * sizeof(float));
        return q;
    };
    w.phi = take(F * B);
    w.sctab = take((size_t)3 * d.D * d.m);  // (D, 2, m) cos / sin of eps B_dj, then (D, m) cos - 1 (bf16x3 forward)
    w.phiTc = take(F * B);
    for (int i = 0; i < d.nlayers - 1; ++i) w.zsave[i] = take((size_t)d.L * HID * B);
    w.jac = take((size_t)B * d.L);
    w.dsc = take((size_t)B * d.L);
    for (int i = 0; i < d.nlayers - 1; ++i) w.dz[i] = take((size_t)d.L * HID * B);
    w.dbase = take((size_t)B * d.L);
    w.dfsc = take((size_t)B * d.L);
    const int S = wgrad_slices(d, B), SS = stream_bwd_slices(d, B);
    w.gpart = (S > 1 || SS > 0) ? take((size_t)(S > SS ? S : SS) * part_layout(d).stride) : nullptr;
    w.w0p = (unsigned short*)take(((size_t)3 * d.L * HID * F + 1) / 2);
    w.whp = (unsigned short*)take(((size_t)(d.nlayers > 2 ? d.nlayers - 2 : 0) * 3 * d.L * HID * HID + 1) / 2);
    w.base_raw = take((size_t)d.L * (1 + 2 * (size_t)d.D) * B);
    w.loss_part = take(33 * (size_t)d.L);
    w.kpart = fwd_kpart_floats(d, B) ? take(fwd_kpart_floats(d, B)) : nullptr;
    w.tickets = w.kpart ? (unsigned*)take((size_t)d.L * (B / BS)) : nullptr;
    w.bytes = off;
    return w;
}

}  // namespace nsvd_pmlp
