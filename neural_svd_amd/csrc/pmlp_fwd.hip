// Fused MFMA path for the headline shapes: every hidden layer 128 wide, B % 32 == 0, F % 32 == 0.
//
// FORWARD  (pmlp_fused_fwd_kernel): one workgroup = one head l x one block of 32 base samples x all
// E = 1+2D stencil points (NC = 32 E sample columns), 4 waves, one per SIMD.  The whole per-head MLP
// chain runs "transposed" - hidden units on the MFMA M axis, samples on the lane (N) axis:
//     Z_i^T[n][c] = sum_k W_i[n][k] * A_{i-1}^T[k][c]
// so a v_mfma_f32_32x32x2_f32 accumulator (column = lane, rows = registers) of layer i is, after
// bias + softplus in registers, exactly the B operand layout of layer i+1's MFMAs; activations only
// cross LDS once per layer (each wave owns 32 of the 128 hidden rows and needs all 128 as K).
//   layer 0: K = F streamed in 32-wide chunks: W_0 tile (128 x 32, rows padded to 36 floats so the
//            ds_read_b128 fragments are bank-conflict free) + phi^T tile (32 x NC) register-staged
//            global -> LDS, double buffered, one barrier per chunk, loads for chunk c+1 in flight
//            under the 80 MFMAs/wave of chunk c;
//   layers 1..: W_i fragments straight from L2 (16 x 16 B per lane), B operand = LDS activations;
//   last layer (128 -> 1): register dot product + cross-wave LDS reduction;
//   epilogue: importance-weighted central-difference Hamiltonian (fd_math.h) -> f, Tf (B, L).
// Grid = (B/32) * L workgroups (256 at hydrogen L=16, B=512: one per CU), remapped so that the
// workgroups of one head share an XCD (its 1 MB W_0 stays in that XCD's L2).
// Reference arithmetic being replaced: examples/models/mlp.py:204-221 x (1+2D) evaluations
// (diff_ops.py:36-45) + diff_ops.py:9-23 + schrodinger/__init__.py:16-22 + examples/__init__.py:7-9.
#include <stdlib.h>
#include <type_traits>
#include "pmlp_common.h"
#include "fd_math.h"

using namespace nsvd_pmlp;

namespace {

struct FwdArgs {
    const float* phiT;   // (B, F) Fourier features of the CENTRE rows, sample-major: [sin(x.B) | cos(x.B)]
    const float* sctab;  // (D, 2, m): cos(eps B_dj), sin(eps B_dj) - the stencil rows are built from the centre
                         // features (even / odd perturbation rows) while the layer-0 tiles are staged
    int m;
    int ldr;
    const float* W[NSVD_MAX_LAYERS];
    const float* b[NSVD_MAX_LAYERS];
    int nlayers;  // weight matrices: nh hidden (128 wide) + the final 128 -> 1
    const float* x;
    const float* scales;
    nsvd_problem prob;
    float log_norm;
    int B, D, L, F;
    float* f;
    float* Tf;
    float* jac;
    float* dsc;
    float* zsave[NSVD_MAX_LAYERS];  // (L, 128, B) per hidden layer, or null: ACTIVATIONS softplus(z) of the centre rows
    const unsigned short* w0p;  // BF3 only: W_0 pre-split into three bf16 planes, fragment-major (w0_split_kernel)
    const unsigned short* whp;  // BF3 only: the hidden layers' W_1 .. likewise
    int plain;      // E = 1 instance only: out = hard_mul_const * base * mask (WaveFunctions.forward), no Hamiltonian;
                    // f receives the output, jac / dsc its derivatives w.r.t. base / scales
    int xcd_remap;  // 0: plain mapping; else HX = number of head groups across the 8 XCDs (1, 2, 4 or 8)
    // split-stencil form (E = 3 instance only): the launch holds `split` = D groups of (B / 32) L workgroups; group g
    // evaluates the centre and the two points x +- eps e_g and writes the RAW head outputs of its two shifted points
    // (group 0: of the centre too, and it alone saves the centre activations) into base_raw (L, ldr) at rows e B + b,
    // stencil order [x, x + eps e_0, x - eps e_0, x + eps e_1, ...]; fd_epilogue_kernel then forms f, Tf. Twice (three
    // times) the workgroups for batches that leave CUs idle, and the only way 7 stencil columns (D = 3) fit at all.
    int split;
    float* base_raw;
    // K-split of layer 0 (KS = 1 / 2 instances, D = 2 stencil; pmlp_common.h: fwd_kslices): the KS = 1 launch
    // holds `ks` x the form's own workgroups, slice k of a workgroup contracts features [k m / ks, (k + 1) m / ks) of the
    // sin block and their cos partners and leaves its accumulators in kpart [slice][workgroup][tile][wave][register][lane];
    // the KS = 2 launch (the split form's own grid) adds the slices in order and runs the rest of the network
    int ks;
    float* kpart;
    int kplain;  // KS = 2, E = 3 only: the partials were left by PLAIN-form workgroups (five tiles per (head, sample block):
                 // centre, even_0, odd_0, even_1, odd_1) - direction g's workgroup adds tiles 0, 1 + 2 g, 2 + 2 g
    // K-split, split form: arrival counters per (head, sample block), or null. The KS = 1 launch zeroes them; in the KS = 2
    // launch the direction group that arrives LAST at a (head, sample block) reads the other groups' raw outputs and forms
    // f, Tf itself (fd_math.h: the arithmetic of fd_epilogue_kernel, same bits) - no epilogue launch (round 6)
    unsigned* tickets;
    unsigned long long* stamps;  // diagnostic build only (NSVD_FWD_STAMPS): per-workgroup s_memtime stamps
};



// 16 bytes per lane global -> LDS without passing through VGPRs (global_load_lds_dwordx4). The LDS
// destination is wave-uniform base + lane * 16; the global source is per lane.
__device__ __forceinline__ void nsvd_glds16(const float* gsrc, float* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

#ifdef NSVD_FWD_STAMPS
#define NSVD_STAMP(i)                                                                   \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();
#else
#define NSVD_STAMP(i)
#endif
#include "pmlp_layer0_bf3.h"

// BF3 = 1: layer 0 on the bf16 MFMA with three-way split operands (nsvd_layer0_bf3 above); everything after layer 0 is
//   the same code (one extra barrier after the K loop: the hidden layers' W tile DMA lands in the stage buffers).
// PL = 1: plain model evaluation (nsvd_model_forward): the E column tiles of a workgroup are E consecutive 32-sample
// tiles of the batch (no stencil, no jets), so that a head's weight tiles are fetched once per 32 E samples
#ifdef NSVD_EO_COUNT
__device__ unsigned long long g_eo_count[8];
#endif
template <int E, int JET, int BF3 = 0, int PL = 0, int KS = 0>
__global__ void __launch_bounds__(256, 1) pmlp_fused_fwd_kernel(FwdArgs a) {
    static_assert(!PL || (!JET && !BF3 && E <= 4), "plain tiles: native fp32 layer 0, at most four sample tiles");
    static_assert(!KS || ((E == 3 || E == 5) && !JET && !BF3 && !PL), "K-split: the D = 2 stencil forms only");
    constexpr int NC = E * BS;
    // Stencil mode (neither jets nor plain tiles), both the native and the bf16x3 kernel: the 2 D shifted evaluations
    // travel through the network in EVEN / ODD form - tile 0 the centre x, tile 1 + 2 d the even part and tile 2 + 2 d the
    // odd part of (value at x + eps e_d, value at x - eps e_d) minus the centre value: z(x +- eps e_d) = z + zE_d +- zO_d.
    // Layer 0 is linear in the features (even rows u (cos d - 1), odd rows +- v sin d: pmlp_layer0_bf3.h), the softplus
    // acts on the triple by its Taylor expansion around the centre, and the epilogue forms the central difference from
    // the even parts (fd_math.h: nsvd_fd_evenodd). The finite-difference Laplacian - what the reference's float32
    // arithmetic, and round 3's kernels, carry with a per-point error of ~|f| at eps = 0.01 - is then never the
    // difference of rounded large numbers: Tf agrees with the float64 stencil to ~1e-5 instead of a few per cent.
    constexpr bool EO = !JET && !PL;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][128][A_LD]   W_0 tile, k contiguous
    float* Bs = smem + 2 * HID * A_LD;      // [2][NC][A_LD]    phi tile (rows = sample columns), k contiguous
    float* Hs = smem;                       // [NC][H_LD]       activations, k contiguous (aliases the stage buffers)
    constexpr int STAGE = 2 * HID * A_LD + 2 * NC * A_LD;
    constexpr int HSZ = NC * H_LD;
    constexpr int STAGE3 = (2 * 2 * 3 * NC * 64) / 4;  // floats: the sample-column planes of two PAIRS of chunks (upper bound)
    constexpr int WL_OFF = BF3 ? HSZ : (STAGE > HSZ ? STAGE : HSZ);
    float* Wl = smem + WL_OFF;   // [4 waves][32][128]  next layer's W rows of each wave (XOR-swizzled chunks)
    constexpr int RED_OFF = BF3 ? (STAGE3 > HSZ + HID * HID ? STAGE3 : HSZ + HID * HID) : WL_OFF + HID * HID;
    float* red = smem + RED_OFF;                      // [4][NC]
    float* outs = red + 4 * NC;                       // [NC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nsb = a.B / (PL ? NC : BS);
    int l, sb, grp = 0, bid = blockIdx.x;
    int kslice = 0;
    if (KS == 1 && a.tickets && blockIdx.x == 0) {
        for (int i = threadIdx.x; i < a.L * (a.B / BS); i += blockDim.x) a.tickets[i] = 0u;
    }
    if (KS == 1) {  // K-split, first launch: which slice of layer 0's contraction
        const int per = nsb * a.L * (a.split > 0 ? a.split : 1);
        kslice = __builtin_amdgcn_readfirstlane(bid / per);
        bid -= kslice * per;
    }
    if (E == 3 && !JET && !BF3 && a.split) {  // split-stencil form: which direction's points this workgroup evaluates
        grp = bid / (nsb * a.L);
        bid -= grp * nsb * a.L;
    }
    xcd_block_map(bid, a.xcd_remap, a.L, nsb, l, sb);  // pmlp_common.h
    // (the map divides by run-time values: computed on the vector ALU although uniform - pin the results to scalar
    // registers so that every address derived from them is scalar too)
    l = __builtin_amdgcn_readfirstlane(l);
    const int b0 = __builtin_amdgcn_readfirstlane(sb * (PL ? NC : BS));
    grp = __builtin_amdgcn_readfirstlane(grp);

    NSVD_STAMP(0)
    // accumulators start from the bias (z = b + W a): its 16 loads fly under the first chunk's staging instead
    // of sitting, exposed, between the K loop and the softplus
    f32x16 acc[E];
    {
        const float* bi0 = a.b[0] + (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float bv = bi0[acc_row(r, hi)];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e][r] = ((JET || EO) && e > 0) ? 0.f : bv;  // (the bias joins tile 0 only)
        }
    }
    // K-split: this workgroup's block of partial accumulators, [tile][wave][register][lane] floats per (slice, workgroup)
    float* kp = nullptr;
    if (KS) {
        const size_t unit = ((size_t)grp * a.L + l) * nsb + sb;
        // [slice][workgroup][tile][wave][lane][16 registers]: a lane's accumulator tile is 64 contiguous bytes
        kp = a.kpart + ((((size_t)kslice * (a.split > 0 ? a.split : 1) * a.L * nsb + unit) * E * 4 + w) * 64 + lane) * 16;
        if (KS == 2 && E == 3 && a.kplain)
            kp = a.kpart + ((((size_t)l * nsb + sb) * 5 * 4 + w) * 64 + lane) * 16;
        if (KS == 1 && kslice > 0) {  // (the bias joins slice 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        }
    }

    if constexpr (BF3) {
        nsvd_layer0_bf3<E, JET>(a, acc, reinterpret_cast<char*>(smem), l, b0);
        __syncthreads();  // the W-tile DMA below lands in the tail of the stage buffers
    } else if constexpr (KS == 2) {
        // K-split, second launch: the slices' partial pre-activations, added in slice order
        const bool kpl = E == 3 && a.kplain;
        const size_t sstride = kpl ? (size_t)a.L * nsb * 5 * 4 * 16 * 64
                                   : (size_t)(a.split > 0 ? a.split : 1) * a.L * nsb * E * 4 * 16 * 64;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int et = (kpl && e > 0) ? 2 * grp + e : e;  // tile index inside the producer's block
            float4 v[4][4];  // [slice][quarter]: every load of a tile requested before the first is used
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    v[k][g] = k < a.ks ? *reinterpret_cast<const float4*>(kp + (size_t)k * sstride + (size_t)et * 4 * 64 * 16 + 4 * g)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t = v[0][g];
#pragma unroll
                for (int k = 1; k < 4; ++k) {  // (slice order; slices beyond ks add zeros)
                    t.x += v[k][g].x; t.y += v[k][g].y; t.z += v[k][g].z; t.w += v[k][g].w;
                }
                acc[e][4 * g] = t.x; acc[e][4 * g + 1] = t.y; acc[e][4 * g + 2] = t.z; acc[e][4 * g + 3] = t.w;
            }
        }
    } else {
    // ------------------------------------------------------------------ layer 0: K = F in chunks of BK
    // Both operands are k-contiguous rows (W_0[l][n][:] and phi[r][:]): each thread moves one float4 of a
    // 32-row slab per step, global -> registers -> LDS; chunk c+1 is in flight while chunk c is multiplied.
    // Named registers, no arrays: hipcc leaves a conditionally written float4 array in scratch.
    const float* W0 = a.W[0] + (size_t)l * HID * a.F;
    const int nch = KS == 1 ? a.F / BK / a.ks : a.F / BK;
    const int kf0 = KS == 1 ? kslice * (a.m / a.ks) : 0;  // first sin feature of this K slice
    // K runs over PAIRS of chunks: 32 sin features k in [32 p, 32 p + 32) and their 32 cos partners m + k. The
    // pair is loaded once (centre row only: 2 float4 per thread + the per-frequency constants) and the E stencil
    // rows of both chunks are generated in registers (even rows u (cos d - 1), odd rows +- v sin d: the angle-addition
    // identities with the centre value taken out),
    //   sin(t +- d) - sin t = sin t (cos d - 1) +- cos t sin d,  cos(t +- d) - cos t = cos t (cos d - 1) -+ sin t sin d,
    // d = eps B_dj (the table holds sin d, cos d and cos d - 1 = -2 sin^2(d / 2)): phi(x +- eps e_d) is never stored.
    // (5x less feature traffic per tile; the feature kernel writes B x F instead of E x B x F.)
    constexpr int DD = PL ? 0 : (JET ? E - 2 : (E - 1) / 2);  // input dimensions
    float4 ra0, ra1, ra2, ra3, rs, rc, cd0, sd0, cd1, sd1, cd2, sd2;
    ra0 = ra1 = ra2 = ra3 = rs = rc = cd0 = sd0 = cd1 = sd1 = cd2 = sd2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s_row = tid >> 3, s_c4 = tid & 7;  // 32 rows x 8 float4 per slab
    // Addresses as UNIFORM base (scalar registers, advanced per chunk by scalar adds) + one 32-bit byte offset per
    // thread, the same for W_0's slabs and the feature rows: the loads take the scalar-base form and the loop carries
    // no 64-bit vector address arithmetic (it did: 10 v_lshl_add_u64 per chunk - and every vector-ALU instruction in
    // this loop costs matrix-pipe time, see the header).
    const unsigned offA = (unsigned)(s_row * a.F + 4 * s_c4) * 4u;
    const unsigned offT = (unsigned)(4 * s_c4) * 4u;
    const float* a_u = W0 + kf0;
    const float* b_u = a.phiT + (size_t)b0 * a.F + kf0;               // centre features, (B, F) row-major
    const float* t_u = a.sctab + (size_t)grp * 2 * a.m + kf0;         // (D, 2, m): this group's direction first
    const float* tc_u = a.sctab + (size_t)(2 * a.D + grp) * a.m + kf0;  // (D, m) behind it: cos(eps B_dj) - 1
    const size_t a_step = (size_t)32 * a.F;
    const int mm = a.m;
#define NSVD_LDGU(ub, off) (*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ub) + (off)))
// chunk c = pair (c >> 1), half (c & 1): HALF = 0 the sin features, 1 their cos partners
#define NSVD_LOAD_CHUNK(c, HALF)                                                 \
    {                                                                            \
        const int kp_ = ((c) >> 1) * BK;                                         \
        const float* pa_ = a_u + ((HALF) ? mm : 0) + kp_;                        \
        ra0 = NSVD_LDGU(pa_, offA);                                              \
        ra1 = NSVD_LDGU(pa_ + a_step, offA);                                     \
        ra2 = NSVD_LDGU(pa_ + 2 * a_step, offA);                                 \
        ra3 = NSVD_LDGU(pa_ + 3 * a_step, offA);                                 \
        if (PL) {       /* this chunk's features of the E sample tiles (rows 32 e + s_row of the block) */ \
            const float* pb_ = b_u + ((HALF) ? mm : 0) + kp_;                    \
            rs = NSVD_LDGU(pb_, offA);                                           \
            if (E > 1) rc = NSVD_LDGU(pb_ + a_step, offA);                       \
            if (E > 2) cd0 = NSVD_LDGU(pb_ + 2 * a_step, offA);                  \
            if (E > 3) sd0 = NSVD_LDGU(pb_ + 3 * a_step, offA);                  \
        } else if (!(HALF)) {                                                    \
            rs = NSVD_LDGU(b_u + kp_, offA);                                     \
            rc = NSVD_LDGU(b_u + mm + kp_, offA);                                \
            /* stencil mode: the cd slots hold cos(eps B) - 1 (even / odd rows); jets: B_dj */ \
            if (DD > 0) cd0 = NSVD_LDGU((EO ? tc_u : t_u) + kp_, offT);          \
            if (DD > 0) sd0 = NSVD_LDGU(t_u + mm + kp_, offT);                   \
            if (DD > 1) cd1 = NSVD_LDGU((EO ? tc_u + mm : t_u + 2 * mm) + kp_, offT); \
            if (DD > 1) sd1 = NSVD_LDGU(t_u + 3 * mm + kp_, offT);               \
            if (DD > 2) cd2 = NSVD_LDGU((EO ? tc_u + 2 * mm : t_u + 4 * mm) + kp_, offT); \
            if (DD > 2) sd2 = NSVD_LDGU(t_u + 5 * mm + kp_, offT);               \
        }                                                                        \
    }
#define NSVD_STS(p, v) (*reinterpret_cast<float4*>(p) = (v))
// (rounds 1-3 generated the shifted rows by angle addition here, u cd +- v sd; measured upper bound for that arithmetic:
// with it removed altogether the cfg2 kernel ran 176.0 us instead of 178.9 - the loop's overhead over its MFMA issue
// is barrier skew and LDS traffic, not the vector ALU. The even / odd rows of round 4 are two multiplies per row.)
// jet rows of a chunk: value u, derivative streams w * b_d (w = the partner feature, sign folded in), Laplacian -q u
#define NSVD_MUL4(o, u, k) o = make_float4(u.x * k.x, u.y * k.y, u.z * k.z, u.w * k.w)
#define NSVD_NMUL4(o, u, k) o = make_float4(-(u.x * k.x), -(u.y * k.y), -(u.z * k.z), -(u.w * k.w))
#define NSVD_STORE_CHUNK(buf, HALF)                                              \
    {                                                                            \
        float* Ab_ = As + (buf) * HID * A_LD + s_row * A_LD + 4 * s_c4;          \
        float* Bb_ = Bs + (buf) * NC * A_LD + s_row * A_LD + 4 * s_c4;           \
        NSVD_STS(Ab_, ra0);                                                      \
        NSVD_STS(Ab_ + 32 * A_LD, ra1);                                          \
        NSVD_STS(Ab_ + 64 * A_LD, ra2);                                          \
        NSVD_STS(Ab_ + 96 * A_LD, ra3);                                          \
        float4 gp_, gm_;                                                         \
        if (PL) {                                                                \
            NSVD_STS(Bb_, rs);                                                   \
            if (E > 1) NSVD_STS(Bb_ + 32 * A_LD, rc);                            \
            if (E > 2) NSVD_STS(Bb_ + 64 * A_LD, cd0);                           \
            if (E > 3) NSVD_STS(Bb_ + 96 * A_LD, sd0);                           \
        } else if (JET) {      /* table: cd_d = B_dj, sd0 = |B_j|^2 */           \
            if (!(HALF)) {  /* sin: d_d = B_d cos, Lap = -|B|^2 sin */           \
                NSVD_STS(Bb_, rs);                                               \
                if (DD > 0) { NSVD_MUL4(gp_, rc, cd0); NSVD_STS(Bb_ + 32 * A_LD, gp_); }   \
                if (DD > 1) { NSVD_MUL4(gp_, rc, cd1); NSVD_STS(Bb_ + 64 * A_LD, gp_); }   \
                if (DD > 2) { NSVD_MUL4(gp_, rc, cd2); NSVD_STS(Bb_ + 96 * A_LD, gp_); }   \
                NSVD_NMUL4(gm_, rs, sd0);                                        \
                NSVD_STS(Bb_ + (DD + 1) * 32 * A_LD, gm_);                       \
            } else {        /* cos: d_d = -B_d sin, Lap = -|B|^2 cos */          \
                NSVD_STS(Bb_, rc);                                               \
                if (DD > 0) { NSVD_NMUL4(gp_, rs, cd0); NSVD_STS(Bb_ + 32 * A_LD, gp_); }  \
                if (DD > 1) { NSVD_NMUL4(gp_, rs, cd1); NSVD_STS(Bb_ + 64 * A_LD, gp_); }  \
                if (DD > 2) { NSVD_NMUL4(gp_, rs, cd2); NSVD_STS(Bb_ + 96 * A_LD, gp_); }  \
                NSVD_NMUL4(gm_, rc, sd0);                                        \
                NSVD_STS(Bb_ + (DD + 1) * 32 * A_LD, gm_);                       \
            }                                                                    \
        } else if (!(HALF)) {  /* sin rows, even / odd: s (cos d - 1), +c sin d  (sin(t +- d) - sin t) */ \
            NSVD_STS(Bb_, rs);                                                   \
            if (DD > 0) { NSVD_MUL4(gp_, rs, cd0); NSVD_MUL4(gm_, rc, sd0); NSVD_STS(Bb_ + 32 * A_LD, gp_); NSVD_STS(Bb_ + 64 * A_LD, gm_); }   \
            if (DD > 1) { NSVD_MUL4(gp_, rs, cd1); NSVD_MUL4(gm_, rc, sd1); NSVD_STS(Bb_ + 96 * A_LD, gp_); NSVD_STS(Bb_ + 128 * A_LD, gm_); }  \
            if (DD > 2) { NSVD_MUL4(gp_, rs, cd2); NSVD_MUL4(gm_, rc, sd2); NSVD_STS(Bb_ + 160 * A_LD, gp_); NSVD_STS(Bb_ + 192 * A_LD, gm_); } \
        } else {        /* cos rows, even / odd: c (cos d - 1), -s sin d  (cos(t +- d) - cos t) */ \
            NSVD_STS(Bb_, rc);                                                   \
            if (DD > 0) { NSVD_MUL4(gp_, rc, cd0); NSVD_NMUL4(gm_, rs, sd0); NSVD_STS(Bb_ + 32 * A_LD, gp_); NSVD_STS(Bb_ + 64 * A_LD, gm_); }   \
            if (DD > 1) { NSVD_MUL4(gp_, rc, cd1); NSVD_NMUL4(gm_, rs, sd1); NSVD_STS(Bb_ + 96 * A_LD, gp_); NSVD_STS(Bb_ + 128 * A_LD, gm_); }  \
            if (DD > 2) { NSVD_MUL4(gp_, rc, cd2); NSVD_NMUL4(gm_, rs, sd2); NSVD_STS(Bb_ + 160 * A_LD, gp_); NSVD_STS(Bb_ + 192 * A_LD, gm_); } \
        }                                                                        \
    }

    // Software pipeline, one barrier per chunk (80 MFMAs per wave between barriers):
    //   * fragments are read one q-group (8 k) ahead of the MFMAs that use them;
    //   * the LAST q-group of chunk c is multiplied AFTER the barrier, under the first fragment reads of
    //     chunk c+1 and the global loads of chunk c+2, so neither latency is exposed;
    //   * chunk c+1 is written to the other LDS buffer in the shadow of chunk c's third q-group.
    NSVD_LOAD_CHUNK(0, 0);
    NSVD_STORE_CHUNK(0, 0);
    __syncthreads();
    NSVD_STAMP(1)
    NSVD_LOAD_CHUNK(1, 1);  // nch = F / 32 is even and >= 4 (F is a multiple of 128)
    Frag<E> f0, f1;
    {
        const float* Ap = As + (32 * w + li) * A_LD + 4 * hi;
        const float* Bp = Bs + li * A_LD + 4 * hi;
        load_frag<E>(f0, Ap, Bp, A_LD);
    }
    // __builtin_amdgcn_sched_barrier(0) fences pin the phase order below; without them hipcc sinks the
    // fragment reads next to their first use and parks the LDS stores right in front of the barrier,
    // which exposes both latencies once per chunk.
#define NSVD_FENCE() __builtin_amdgcn_sched_barrier(0)
// n x { 1 MFMA, 1 instruction of class `mask` } in the current scheduling region (LLVM SchedGroupMask:
// 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write)
#define NSVD_INTERLEAVE(n, mask)                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         \
        __builtin_amdgcn_sched_group_barrier((mask), 1, 0);        \
    }
    // each region = 4E MFMAs + the memory instructions of the NEXT stage, interleaved one per MFMA gap
    // (sched_group_barrier): a memory instruction issued in the shadow of an executing MFMA is free, a
    // block of 9..15 of them in a row stalls the matrix pipe for ~250..500 cycles per chunk.
    // The steady-state body is branch free (the last two chunks are peeled) so every region is one
    // basic block the scheduler can interleave.
// PAR = parity of c: chunk c+1 (stored here) is the other half of a pair, chunk c+2 (loaded here) the same half
#define NSVD_CHUNK_BODY(c, DO_STORE, DO_LOAD, PAR)                                              \
    {                                                                                           \
        const int cur = (c) & 1;                                                                \
        const float* Ap = As + cur * HID * A_LD + (32 * w + li) * A_LD + 4 * hi;                \
        const float* Bp = Bs + cur * NC * A_LD + li * A_LD + 4 * hi;                            \
        load_frag<E>(f1, Ap + 8, Bp + 8, A_LD);                                                 \
        mma_frag<E>(acc, f0);                                                                   \
        NSVD_INTERLEAVE(1 + E, 0x100);                                                          \
        NSVD_FENCE();                                                                           \
        load_frag<E>(f0, Ap + 16, Bp + 16, A_LD);                                               \
        mma_frag<E>(acc, f1);                                                                   \
        NSVD_INTERLEAVE(1 + E, 0x100);                                                          \
        NSVD_FENCE();                                                                           \
        load_frag<E>(f1, Ap + 24, Bp + 24, A_LD);                                               \
        if (DO_STORE) NSVD_STORE_CHUNK(cur ^ 1, 1 - (PAR));                                     \
        mma_frag<E>(acc, f0);                                                                   \
        NSVD_INTERLEAVE(1 + E, 0x100);                                                          \
        if (DO_STORE) NSVD_INTERLEAVE(4 + E, 0x200);                                            \
        NSVD_FENCE();                                                                           \
        __syncthreads();                                                                        \
        if (DO_STORE) {                                                                         \
            const float* An = As + (cur ^ 1) * HID * A_LD + (32 * w + li) * A_LD + 4 * hi;      \
            const float* Bn = Bs + (cur ^ 1) * NC * A_LD + li * A_LD + 4 * hi;                  \
            load_frag<E>(f0, An, Bn, A_LD);                                                     \
        }                                                                                       \
        if (DO_LOAD) NSVD_LOAD_CHUNK((c) + 2, (PAR));                                           \
        mma_frag<E>(acc, f1);                                                                   \
        if (DO_STORE) NSVD_INTERLEAVE(1 + E, 0x100);                                            \
        if (DO_LOAD) NSVD_INTERLEAVE(PL ? 4 + E : ((PAR) ? 4 : 6 + 2 * DD), 0x020);             \
        NSVD_FENCE();                                                                           \
    }
    {
        int c = 0;
        for (; c + 2 < nch; c += 2) {
            NSVD_CHUNK_BODY(c, true, true, 0)
            NSVD_CHUNK_BODY(c + 1, true, true, 1)
        }
        NSVD_CHUNK_BODY(c, true, false, 0)      // c == nch - 2
        NSVD_CHUNK_BODY(c + 1, false, false, 1)
    }
#undef NSVD_CHUNK_BODY
#undef NSVD_LOAD_CHUNK
#undef NSVD_STORE_CHUNK
#undef NSVD_PM
#undef NSVD_MUL4
#undef NSVD_NMUL4
#undef NSVD_LDGU
#undef NSVD_STS

    }
    NSVD_STAMP(2)
    if constexpr (KS == 1) {  // K-split, first launch: leave this slice's accumulators
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(kp + (size_t)e * 4 * 64 * 16 + 4 * g) =
                    make_float4(acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]);
        return;
    }
    // ------------------------------------------------------------------ hidden layers 1 .. nh-1
    const int nh = a.nlayers - 1;
    for (int i = 0; i < nh; ++i) {
        // next layer's weights for this wave (rows 32w..32w+31, all 128 columns) go global -> LDS by LDS-DMA
        // (global_load_lds_dwordx4: no VGPRs, no ds_write), issued before the softplus so they land under
        // it. One instruction moves 2 rows x 512 B; the tile is unpadded, so the 16-B chunks of row r are
        // stored XOR-swizzled by (r & 15) - applied here on the SOURCE address, and again on the fragment
        // reads below - which makes the ds_read_b128 fragment reads bank-conflict free.
        // (Per-lane fragment loads straight from global touch 64 cache lines per instruction: 9k cycles.)
        const bool has_next = (i + 1 < nh);
        float* Wt = Wl + w * 32 * HID;
        // order matters for the wait counters (vmcnt retires in issue order): the loads of the next layer's bias
        // (or of the last layer's weights) are issued BEFORE the DMA, the pre-activation stores AFTER it, so that
        // "vmcnt(#stores)" below means "the DMA has landed" without draining the stores. All of them land under
        // the softplus.
        float nb[16];  // next layer's bias rows of this lane, or (last hidden layer) the 128 -> 1 weights
        {
            const float* src = (has_next ? a.b[i + 1] : a.W[nh]) + (size_t)l * HID + 32 * w;
#pragma unroll
            for (int r = 0; r < 16; ++r) nb[r] = src[acc_row(r, hi)];
        }
        // BF3: the next layer on the bf16 MFMA as well (three-way split operands, six partial products: the arithmetic
        // of layer 0). A wave multiplies its own 32 rows of W_{i+1} only: its A fragments - 8 k-steps x 3 planes, pre-split
        // fragment-major by w0_split_kernel - come straight from global into registers, requested here so that they land
        // under the softplus; no weight tile in LDS, no DMA.
        uint4 wa[8][3];
        if constexpr (BF3) {
            if (has_next) {
                const char* wp = reinterpret_cast<const char*>(a.whp) +
                                 ((((size_t)i * a.L + l) * 4 + w) * (8 * 3 * 64) + lane) * 16;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                    for (int p = 0; p < 3; ++p) wa[ks][p] = *reinterpret_cast<const uint4*>(wp + (ks * 3 + p) * 1024);
            }
        }
        if (!BF3 && has_next) {
            const float* Wn = a.W[i + 1] + ((size_t)l * HID + 32 * w + hi) * HID;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = 2 * j + hi;
                nsvd_glds16(Wn + (size_t)(2 * j) * HID + 4 * (li ^ (row & 15)), Wt + 2 * j * HID);
            }
        }
        // softplus in registers; the centre rows' ACTIVATIONS are saved for the backward (its kernels then need
        // no softplus: sigmoid(z) = 1 - exp(-softplus(z)), and the weight gradients contract activations)
        float* zs = (a.zsave[i] && grp == 0) ? a.zsave[i] + ((size_t)l * HID + 32 * w) * a.B + b0 + li : nullptr;
        if (EO) {
            // stencil mode: tile 0 holds z(x), tiles 1 + 2 d / 2 + 2 d the even / odd parts zE, zO of
            // z(x +- eps e_d) - z(x) = zE +- zO. The softplus acts on the triple by its Taylor expansion around z
            // to sixth order (s = sigmoid z, p = s (1 - s): c1 = s, c2 = p / 2, c3 = p (1 - 2 s) / 6, c4 = p (1 - 6 p) / 24,
            // c5 = p (1 - 2 s)(1 - 12 p) / 120, c6 = p (1 - 30 p + 120 p^2) / 720), with zO = O(delta), zE = O(delta^2), w = zO^2:
            //   even' = c1 zE + c2 (zE^2 + w) + c3 zE (zE^2 + 3 w) + c4 w (w + 6 zE^2) + 5 c5 zE w^2 + c6 w^3   + O(delta^8)
            //   odd'  = zO [c1 + 2 c2 zE + c3 (3 zE^2 + w) + 4 c4 zE w + c5 w^2]                               + O(delta^7)
            // - delta is 2^-7 .. 2^-4 at the reference's initialisations and stays below ~0.3 in trained models: the
            // truncation is delta^6 of the signal (measured with W_0 scaled by 4 at configs[2], delta ~ 0.25: the
            // fourth-order form was 3e-4 from the float64 stencil). No transcendental for the shifted tiles.
            // The expansion wants small perturbations - what eps = 0.01 gives; laplacian_eps and the weights are the
            // caller's, though: a LANE (sample) with a perturbation beyond NSVD_EO_TAYLOR_MAX among the four accumulator
            // registers of a group takes softplus(z0 + d) - softplus(z0) = log1p(s expm1(d)) for that group instead
            // (nsvd_softplus_evenodd_large: the even part then loses log2(1 / delta) bits only). One v_max3 per pair of squares and one skipped branch
            // per group when no lane needs it (the scripts' settings: delta < 0.1, tails to ~0.3 at configs[2]); the
            // decision is the sample's own, so a sample still never sees its neighbours.
            constexpr int DDE = (E - 1) / 2;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float zEv[4][DDE + 1], zOv[4][DDE + 1], evv[4][DDE + 1], odv[4][DDE + 1], z0v[4], s0v[4];  // (+ 1: E = 1)
                float big = 0.f;  // max of zO^2, zE^2 over the group
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * g4 + jj;
                    const float z0 = acc[0][r];
                    // sigmoid without the threshold select: above z = 20 e^-z < 2.1e-9 vanishes in 1 + t, so s = 1 and
                    // p = s (1 - s) = 0 exactly - what torch's softplus threshold prescribes - with no compare
                    const float et = __builtin_amdgcn_exp2f(-fabsf(z0) * NSVD_LOG2E);
                    const float rt = __builtin_amdgcn_rcpf(1.0f + et);
                    const float s1 = z0 >= 0.0f ? rt : et * rt;
                    const float sq = s1 * (1.f - s1);
                    const float t12 = fmaf(-2.f, s1, 1.f);
                    const float c2 = 0.5f * sq;
                    const float c3 = sq * t12 * (1.f / 6.f);
                    const float c4 = sq * fmaf(-6.f, sq, 1.f) * (1.f / 24.f);
                    const float c5 = c3 * fmaf(-12.f, sq, 1.f) * (1.f / 20.f);
                    const float c6 = sq * fmaf(sq, fmaf(120.f, sq, -30.f), 1.f) * (1.f / 720.f);
#pragma unroll
                    for (int d = 0; d < DDE; ++d) {
                        const float zE = acc[1 + 2 * d][r], zO = acc[2 + 2 * d][r];
                        const float w = zO * zO, e2 = zE * zE;
                        zEv[jj][d] = zE; zOv[jj][d] = zO;
                        big = fmaxf(big, fmaxf(w, e2));
                        float ev = fmaf(c6, w, 5.f * c5 * zE);                   // (c6 w + 5 c5 zE) w^2
                        ev = fmaf(ev, w, c4 * fmaf(6.f, e2, w));                 // + c4 (w + 6 zE^2), all x w
                        ev = fmaf(ev, w, c3 * zE * fmaf(3.f, w, e2));            // + c3 zE (zE^2 + 3 w)
                        ev = fmaf(c2, e2 + w, ev);                               // + c2 (zE^2 + w)
                        float od = fmaf(c5, w, 4.f * c4 * zE);                   // (c5 w + 4 c4 zE) w
                        od = fmaf(od, w, c3 * fmaf(3.f, e2, w));                 // + c3 (3 zE^2 + w)
                        od = fmaf(2.f * c2, zE, od) + s1;
                        evv[jj][d] = fmaf(s1, zE, ev);
                        odv[jj][d] = zO * od;
                    }
                    z0v[jj] = z0;
                    s0v[jj] = nsvd_softplus(z0);
                }
                if (__builtin_expect(big > NSVD_EO_TAYLOR_MAX * NSVD_EO_TAYLOR_MAX, 0)) {
#ifdef NSVD_EO_COUNT
                    atomicAdd(&g_eo_count[i], 1ull);
#endif
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int d = 0; d < DDE; ++d) {
                            float evl, odl;
                            nsvd_softplus_evenodd_large(z0v[jj], zEv[jj][d], zOv[jj][d], &evl, &odl);
                            const bool lg = fmaxf(fabsf(zOv[jj][d]), fabsf(zEv[jj][d])) > NSVD_EO_TAYLOR_MAX;
                            evv[jj][d] = lg ? evl : evv[jj][d];   // (a small pair of a flagged sample keeps its expansion)
                            odv[jj][d] = lg ? odl : odv[jj][d];
                        }
                }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int r = 4 * g4 + jj;
#pragma unroll
                    for (int d = 0; d < DDE; ++d) {
                        acc[1 + 2 * d][r] = evv[jj][d];
                        acc[2 + 2 * d][r] = odv[jj][d];
                    }
                    acc[0][r] = s0v[jj];
                }
            }
            if (zs) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zs[(size_t)acc_row(r, hi) * a.B] = acc[0][r];
            }
        } else if (JET) {
            // forward-mode jet through the softplus, all streams of a (row, sample) in this lane's registers
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z0 = acc[0][r];
                const float s1 = nsvd_sigmoid(z0);
                const float s2 = z0 > NSVD_SOFTPLUS_THRESHOLD ? 0.f : s1 * (1.f - s1);
                float q = 0.f;
#pragma unroll
                for (int e = 1; e < E - 1; ++e) {
                    q = fmaf(acc[e][r], acc[e][r], q);
                    acc[e][r] *= s1;
                }
                acc[E - 1][r] = fmaf(s1, acc[E - 1][r], s2 * q);
                acc[0][r] = nsvd_softplus(z0);
            }
            if (zs) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zs[(size_t)acc_row(r, hi) * a.B] = acc[0][r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = nsvd_softplus(acc[0][r]);
            if (zs) {
#pragma unroll
                for (int r = 0; r < 16; ++r) zs[(size_t)acc_row(r, hi) * a.B] = acc[0][r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int e = 1; e < E; ++e) acc[e][r] = nsvd_softplus(acc[e][r]);
            if (PL && zs) {  // every tile is a tile of samples: all of them are saved
#pragma unroll
                for (int e = 1; e < E; ++e)
#pragma unroll
                    for (int r = 0; r < 16; ++r) zs[(size_t)acc_row(r, hi) * a.B + e * BS] = acc[e][r];
            }
        }
        NSVD_STAMP(3 + 4 * i)
        if (!has_next) {
            // ---------------------------------------------------------- last layer 128 -> 1 (weights in nb)
            float part[E];
#pragma unroll
            for (int e = 0; e < E; ++e) part[e] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int e = 0; e < E; ++e) part[e] = fmaf(nb[r], acc[e][r], part[e]);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                part[e] += __shfl_xor(part[e], 32, 64);
                if (hi == 0) red[w * NC + e * BS + li] = part[e];
            }
            break;
        }
        NSVD_STAMP(4 + 4 * i)
        if constexpr (BF3) {
            // activations -> bf16 planes in LDS, [plane][column][128 k + 16 B pad] (272-B rows: the ds_read_b128
            // fragment reads of 16 consecutive columns hit 16 distinct 4-bank groups); registers 4g .. 4g+3 of a lane
            // are 4 consecutive k = 32 w + 8 g + 4 hi + (0..3): one 8-byte store per plane.
            // Stencil mode (as in layer 0, pmlp_layer0_bf3.h): tiles 1 .. E-1 hold the even / odd perturbations of the
            // activations - small, so two planes and three partial products carry them; the centre tile takes all six
            // (on two alternating accumulators).
            constexpr bool DELTA = !JET;
            constexpr int HB_ROW = 2 * HID + 16, HB_PL = NC * HB_ROW;  // bytes (plane 2: the centre tile's 32 rows only
                                                                       // in stencil mode - the same stride is kept)
            char* Hb = reinterpret_cast<char*>(smem);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // previous LDS contents are dead
#pragma unroll
            for (int e = 0; e < E; ++e) {
                char* hcol = Hb + (e * BS + li) * HB_ROW + 2 * (32 * w + 4 * hi);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (DELTA && e > 0) {
                        uint2 p0, p1;
                        nsvd_bf2_split(make_float4(acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]),
                                       p0, p1);
                        *reinterpret_cast<uint2*>(hcol + 16 * g) = p0;
                        *reinterpret_cast<uint2*>(hcol + 16 * g + HB_PL) = p1;
                    } else {
                        uint2 p0, p1, p2;
                        nsvd_bf3_split(make_float4(acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]),
                                       p0, p1, p2);
                        *reinterpret_cast<uint2*>(hcol + 16 * g) = p0;
                        *reinterpret_cast<uint2*>(hcol + 16 * g + HB_PL) = p1;
                        *reinterpret_cast<uint2*>(hcol + 16 * g + 2 * HB_PL) = p2;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            NSVD_STAMP(5 + 4 * i)
            f32x16 accb;  // stencil mode: the centre tile's second accumulator
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                accb[r] = 0.f;
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e][r] = e > 0 ? 0.f : nb[r];  // (jets: the bias joins the value stream only)
            }
            // K = 128 in 8 k-steps of 16: fragments of k-step ks + 1 are read under the MFMAs of k-step ks
            const char* Bp = Hb + li * HB_ROW + 16 * hi;
            nsvd_bf16x8 hb[2][E][3];
            auto hfrags = [&](int buf, int ks) {
#pragma unroll
                for (int e = 0; e < E; ++e)
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        if (p < 2 || !DELTA || e == 0)
                            hb[buf][e][p] = *reinterpret_cast<const nsvd_bf16x8*>(Bp + p * HB_PL + e * BS * HB_ROW + 32 * ks);
            };
            hfrags(0, 0);
            constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};  // (A plane, B plane), smallest first
            constexpr int DA[3] = {1, 0, 0}, DB[3] = {0, 1, 0};                    // the perturbation tiles' three products
            constexpr int NFRH = DELTA ? 3 + 2 * (E - 1) : 3 * E, NMMH = DELTA ? 6 + 3 * (E - 1) : 6 * E;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 1 < 8) hfrags((ks + 1) & 1, ks + 1);
                auto A = [&](int p) { return __builtin_bit_cast(nsvd_bf16x8, wa[ks][p]); };
                if constexpr (DELTA) {
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        if (t & 1) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[t]), hb[ks & 1][0][TB[t]], accb, 0, 0, 0);
                        else acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[t]), hb[ks & 1][0][TB[t]], acc[0], 0, 0, 0);
#pragma unroll
                        for (int e = 1; e < E; ++e)
                            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(DA[t]), hb[ks & 1][e][DB[t]], acc[e], 0, 0, 0);
                    }
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[3]), hb[ks & 1][0][TB[3]], accb, 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[4]), hb[ks & 1][0][TB[4]], acc[0], 0, 0, 0);
                    accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[5]), hb[ks & 1][0][TB[5]], accb, 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[t]), hb[ks & 1][e][TB[t]], acc[e], 0, 0, 0);
                    }
                }
                // the next k-step's fragment reads spread under this k-step's MFMAs
                if (ks + 1 < 8) {
#pragma unroll
                    for (int q = 0; q < NFRH; ++q) {
                        if (q < NMMH - NFRH) __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                NSVD_FENCE();
            }
            if constexpr (DELTA) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] += accb[r];  // (the even / odd tiles stay apart)
            }
            NSVD_STAMP(6 + 4 * i)
            continue;
        }
        // raw barriers: __syncthreads() would drain the 16 stores above (vmcnt(0)) before every barrier
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // previous LDS contents are dead
        // registers 4g..4g+3 of a lane are 4 consecutive hidden rows 8g + 4hi + (0..3): one 16-B store
        // into the [column][k] image
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float* hcol = Hs + (e * BS + li) * H_LD + 32 * w + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(hcol + 8 * g) =
                    make_float4(acc[e][4 * g], acc[e][4 * g + 1], acc[e][4 * g + 2], acc[e][4 * g + 3]);
        }
        // DMA done, the 16 stores may still fly (16 E of them with plain tiles: beyond the 6-bit counter from E = 4)
        if (zs && !(PL && E > 1)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        NSVD_STAMP(5 + 4 * i)
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = ((JET || EO) && e > 0) ? 0.f : nb[r];
        // K = 128 in 16 q-groups, fragments read one q-group ahead, one LDS read per MFMA gap
        const float* Ap = Wt + li * HID;        // + 4 * ((2q + hi) ^ (li & 15)): swizzled 16-B chunk
        const int sw = li & 15;
        const float* Bp = Hs + li * H_LD + 4 * hi;
        Frag<E> g0, g1;
        load_frag<E>(g0, Ap + 4 * (hi ^ sw), Bp, H_LD);
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            load_frag<E>(g1, Ap + 4 * ((2 * (q + 1) + hi) ^ sw), Bp + 8 * (q + 1), H_LD);
            mma_frag<E>(acc, g0);
            NSVD_INTERLEAVE(1 + E, 0x100);
            NSVD_FENCE();
            if (q + 2 < 16) load_frag<E>(g0, Ap + 4 * ((2 * (q + 2) + hi) ^ sw), Bp + 8 * (q + 2), H_LD);
            mma_frag<E>(acc, g1);
            if (q + 2 < 16) NSVD_INTERLEAVE(1 + E, 0x100);
            NSVD_FENCE();
        }
        NSVD_STAMP(6 + 4 * i)
    }

#undef NSVD_INTERLEAVE
#undef NSVD_FENCE
    NSVD_STAMP(12)
    __syncthreads();
    if (E == 3 && !JET && !BF3 && a.split) {
        // split-stencil form: raw outputs of the 128 -> 1 layer, one thread per (point, sample)
        if (tid < NC) {
            const int e_t = tid / BS, sidx = tid - e_t * BS;
            if (e_t > 0 || grp == 0) {
                // (even / odd form: rows 1 + 2 g, 2 + 2 g receive the even / odd perturbation of direction g; the bias
                // of the 128 -> 1 layer joins the centre row only)
                const float bve = (red[tid] + red[NC + tid]) + (red[2 * NC + tid] + red[3 * NC + tid]) +
                                  (e_t == 0 ? a.b[nh][l] : 0.f);
                const int eg = e_t == 0 ? 0 : 2 * grp + e_t;  // x + eps e_g at 1 + 2 g, x - eps e_g at 2 + 2 g
                a.base_raw[(size_t)l * a.ldr + (size_t)eg * a.B + b0 + sidx] = bve;
            }
        }
        if constexpr (KS == 2) if (a.tickets) {
            // the last of the `split` direction groups to arrive at this (head, sample block) forms f, Tf: every wave's
            // stores are out, ONE agent-scope release, the ticket; the last arriver acquires once and reads with plain
            // loads (cdna_hip_programming.md, in-launch split-K reduction: this order, always). Correct wherever the
            // groups run (other XCDs included); which group arrives last changes nothing in the arithmetic.
            // (the flag lives in the reduction scratch `red`, free behind the barrier: no second LDS object - a new
            // __shared__ variable would be allocated in EVERY instance of this template, the headline's 155 KB one included)
            int& last_s = *reinterpret_cast<int*>(red);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned old = __hip_atomic_fetch_add(a.tickets + (size_t)l * nsb + sb, 1u, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT);
                last_s = old == (unsigned)a.split - 1u;
                if (last_s) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __syncthreads();
            if (last_s && tid < BS) {
                const int b = b0 + tid;
                float xc[NSVD_FD_MAXD], bE[NSVD_FD_MAXD], bO[NSVD_FD_MAXD];
                for (int dd = 0; dd < a.D; ++dd) xc[dd] = a.x[(size_t)b * a.D + dd];
                const float* br = a.base_raw + (size_t)l * a.ldr + b;
                const float b0v = __builtin_nontemporal_load(br);
                for (int dd = 0; dd < a.D; ++dd) {
                    bE[dd] = __builtin_nontemporal_load(br + (size_t)(1 + 2 * dd) * a.B);
                    bO[dd] = __builtin_nontemporal_load(br + (size_t)(2 + 2 * dd) * a.B);
                }
                const float s_l = a.scales ? a.scales[l] : 0.f;
                const NsvdFdOut o = nsvd_fd_evenodd(b0v, bE, bO, xc, a.D, a.scales != nullptr, s_l, a.prob, a.log_norm);
                const size_t idx = (size_t)b * a.L + l;
                a.f[idx] = o.f;
                a.Tf[idx] = o.Tf;
                if (a.jac) a.jac[idx] = o.jac;
                if (a.dsc) a.dsc[idx] = o.dsc;
            }
        }
        return;
    }
    // ------------------------------------------------------------------ FD Hamiltonian epilogue
    // one thread per (stencil point, sample): output of the 128 -> 1 layer, then g_e = sqrt p(x_e) c base_e mask(x_e)
    // (the exp / sqrt heavy part, E x 32 threads wide instead of a 5-point loop on 32 threads)
    float* gs = outs;         // [NC]   head outputs (stencil mode: centre, even / odd perturbations; jets: the streams)
    if (PL) {
        // model(x) = c * base * exp(-|x| / scales_l)   (reference pde/__init__.py:15-16), any input dimension
        if (tid < NC) {
            const int b = b0 + tid;
            const float bv = (red[tid] + red[NC + tid]) + (red[2 * NC + tid] + red[3 * NC + tid]) + a.b[nh][l];
            const float c = a.prob.hard_mul_const;
            float mk = 1.f, r = 0.f, s_l = 1.f;
            if (a.scales) {
                float r2 = 0.f;
                for (int d = 0; d < a.D; ++d) {
                    const float xv = a.x[(size_t)b * a.D + d];
                    r2 = fmaf(xv, xv, r2);
                }
                r = sqrtf(r2);
                s_l = a.scales[l];
                mk = expf(-r / s_l);
            }
            const size_t idx = (size_t)b * a.L + l;
            a.f[idx] = c * bv * mk;
            if (a.jac) a.jac[idx] = c * mk;
            if (a.dsc) a.dsc[idx] = a.scales ? c * bv * mk * r / (s_l * s_l) : 0.f;
        }
        NSVD_STAMP(14)
        return;
    }
    if (JET) {
        // streams of the 128 -> 1 layer (its bias joins the value stream), then the closed-form product rule
        if (tid < NC)
            gs[tid] = (red[tid] + red[NC + tid]) + (red[2 * NC + tid] + red[3 * NC + tid]) + (tid < BS ? a.b[nh][l] : 0.f);
        __syncthreads();
        if (tid < BS) {
            const int b = b0 + tid;
            float xc[NSVD_FD_MAXD], db[NSVD_FD_MAXD];
            for (int d = 0; d < a.D; ++d) {
                xc[d] = a.x[(size_t)b * a.D + d];
                db[d] = gs[(1 + d) * BS + tid];
            }
            const float s_l = a.scales ? a.scales[l] : 0.f;
            const NsvdFdOut o = nsvd_fd_exact(gs[tid], db, gs[(E - 1) * BS + tid], xc, a.D, a.scales != nullptr, s_l,
                                              a.prob, a.log_norm);
            const size_t idx = (size_t)b * a.L + l;
            a.f[idx] = o.f;
            a.Tf[idx] = o.Tf;
            if (a.jac) a.jac[idx] = o.jac;
            if (a.dsc) a.dsc[idx] = o.dsc;
        }
        NSVD_STAMP(14)
        return;
    }
    {
        // stencil mode: outputs of the 128 -> 1 layer in even / odd form (its bias joins the centre)
        if (tid < NC)
            gs[tid] = (red[tid] + red[NC + tid]) + (red[2 * NC + tid] + red[3 * NC + tid]) + (tid < BS ? a.b[nh][l] : 0.f);
        __syncthreads();
        if (tid < BS) {
            const int b = b0 + tid;
            float xc[NSVD_FD_MAXD], bE[NSVD_FD_MAXD], bO[NSVD_FD_MAXD];
            for (int d = 0; d < a.D; ++d) {
                xc[d] = a.x[(size_t)b * a.D + d];
                bE[d] = gs[(1 + 2 * d) * BS + tid];
                bO[d] = gs[(2 + 2 * d) * BS + tid];
            }
            const float s_l = a.scales ? a.scales[l] : 0.f;
            const NsvdFdOut o = nsvd_fd_evenodd(gs[tid], bE, bO, xc, a.D, a.scales != nullptr, s_l, a.prob, a.log_norm);
            const size_t idx = (size_t)b * a.L + l;
            a.f[idx] = o.f;
            a.Tf[idx] = o.Tf;
            if (a.jac) a.jac[idx] = o.jac;
            if (a.dsc) a.dsc[idx] = o.dsc;
        }
        NSVD_STAMP(14)
        return;
    }
}

#include "pmlp_plain_fwd.h"

template <int E, int BF3 = 0>
size_t fwd_lds_bytes() {
    constexpr int NC = E * BS;
    constexpr int STAGE = 2 * HID * A_LD + 2 * NC * A_LD;
    constexpr int HSZ = NC * H_LD;
    constexpr int STAGE3 = (2 * 2 * 3 * NC * 64) / 4;
    constexpr int RED_OFF = BF3 ? (STAGE3 > HSZ + HID * HID ? STAGE3 : HSZ + HID * HID)
                                : (STAGE > HSZ ? STAGE : HSZ) + HID * HID;
    return (RED_OFF + 5 * NC) * sizeof(float);
}

template <int E, int JET = 0, int BF3 = 0, int PL = 0, int KS = 0>
int launch_fwd(const FwdArgs& a, hipStream_t s, int prof = 3) {  // prof: bit 0 / 1 = this launch opens / closes the bracket
    const size_t lds = fwd_lds_bytes<E, BF3>();
    static bool attr_done = false;  // idempotent, racing threads set the same value
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)pmlp_fused_fwd_kernel<E, JET, BF3, PL, KS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_done = true;
    }
    const int grid = (a.B / (PL ? E * BS : BS)) * a.L * (a.split > 0 ? a.split : 1) * (KS == 1 ? a.ks : 1);
    if (prof & 1) nsvd_prof_begin(s);
    hipLaunchKernelGGL((pmlp_fused_fwd_kernel<E, JET, BF3, PL, KS>), dim3(grid), dim3(256), lds, s, a);
    if (prof & 2) nsvd_prof_end(s);
    NSVD_CHECK_LAUNCH();
    return 0;
}
}  // namespace

// shape conditions shared by the operator path and the plain-model path (both end in the same fused backward)
static bool fused_shape_ok(const nsvd_model_desc& d, int B) {
    if (d.nlayers < 2) return false;
    for (int i = 0; i < d.nlayers - 1; ++i)
        if (d.dims[i] != HID) return false;
    if (B % BS != 0 || B > 65536) return false;
    if (B / wgrad_slices(d, B) > 8192) return false;  // one head's slice of dbase is staged in LDS (wgrad C)
    if ((2 * d.m) % HID != 0) return false;  // layer-0 weight gradient uses 128-wide feature tiles
    // the weight-gradient epilogue addresses every tensor with 32-bit byte offsets (W_0 is the largest)
    if ((size_t)d.L * HID * (size_t)(2 * d.m) * sizeof(float) >= ((size_t)1 << 32)) return false;
    return true;
}

bool nsvd_fused_supported(const nsvd_model_desc& d, int B, bool exact) {
    // E <= 5 columns per sample: the forward's LDS image (155 KB at E = 5). Stencil: E = 1 + 2D columns for D <= 2,
    // and the split form (one direction's two points + the centre per workgroup, FwdArgs::split) for D = 3; the
    // exact-Laplacian jets have E = D + 2, so D <= 3.
    (void)exact;
    if (d.D < 1 || d.D > 3) return false;
    return fused_shape_ok(d, B);
}

size_t nsvd_fused_workspace_bytes(const nsvd_model_desc& d, int B) { return carve_fused(d, B, nullptr).bytes; }

int nsvd_fused_features(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                        int B, void* ws, int save, hipStream_t s, const NsvdSampler* sampler, float* xout) {
    const FusedWs w = carve_fused(d, B, ws);
    return nsvd_fourier_stencil(x, p.fourier_B, w.phi, save ? w.phiTc : nullptr, w.sctab, B, d.D, d.m, prob.eps,
                                sampler, xout, s);
}

int nsvd_fused_forward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                       int B, float* f, float* Tf, void* ws, int save, hipStream_t s, int bf3) {
    const FusedWs w = carve_fused(d, B, ws);
    const int E = 1 + 2 * d.D, R = E * B, F = 2 * d.m;
    int rc = 0;
    if (!(save & 2)) {  // bit 1 of `save`: the features are already in the workspace (nsvd_fused_features)
        rc = nsvd_fused_features(d, p, prob, x, B, ws, save & 1, s);
        if (rc) return rc;
    }
    const bool planes_ready = (save & 4) != 0;  // bf16x3: the weight planes in this workspace are current (nsvd.h)
    save &= 1;
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.phiT = w.phi;
    a.sctab = w.sctab;
    a.m = d.m;
    a.ldr = R;
    a.nlayers = d.nlayers;
    for (int i = 0; i < d.nlayers; ++i) {
        a.W[i] = p.W[i];
        a.b[i] = p.b[i];
        a.zsave[i] = (save && i < d.nlayers - 1) ? w.zsave[i] : nullptr;
    }
    a.x = x;
    a.scales = d.has_exp_mask ? p.scales : nullptr;
    a.prob = prob;
    a.log_norm = nsvd_gauss_log_norm(d.D, prob.sigma);
    a.B = B; a.D = d.D; a.L = d.L; a.F = F;
    a.f = f; a.Tf = Tf;
    a.jac = save ? w.jac : nullptr;
    a.dsc = (save && d.has_exp_mask) ? w.dsc : nullptr;
    // XCD-aware block mapping: split heads into HX groups and sample blocks into 8/HX groups, minimising the
    // bytes each XCD pulls through its L2: (L/HX) * |W_0 slab| + (nsb/SX) * |phi slab|
    a.xcd_remap = pick_xcd_remap(d.L, B / BS, F);
#ifdef NSVD_FWD_STAMPS
    a.stamps = (unsigned long long*)w.dz[0];  // diagnostic build: stamps land in the (then unused) dz_0 scratch
#endif
    if (bf3) {  // opt-in: layer 0 on the bf16 MFMA with three-way split operands
        W0SplitArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.W = reinterpret_cast<const float4*>(p.W[0]);
        sa.P = reinterpret_cast<uint4*>(w.w0p);
        sa.nhid = d.nlayers - 2;
        for (int i = 0; i < sa.nhid; ++i) sa.Wh[i] = reinterpret_cast<const float4*>(p.W[i + 1]);
        sa.Ph = reinterpret_cast<uint4*>(w.whp);
        sa.L = d.L; sa.m = d.m;
        if (!planes_ready) {  // (a fused bf16x3 step leaves the planes of the weights it has just updated)
            hipLaunchKernelGGL(w0_split_kernel, dim3(2048), dim3(256), 0, s, sa);
            NSVD_CHECK_LAUNCH();
        }
        a.w0p = w.w0p;
        a.whp = w.whp;
        if (prob.eps <= 0.f) {
            switch (d.D) {
                case 1: return launch_fwd<3, 1, 1>(a, s);
                case 2: return launch_fwd<4, 1, 1>(a, s);
                case 3: return launch_fwd<5, 1, 1>(a, s);
            }
            return NSVD_EUNSUPPORTED;
        }
        switch (E) {
            case 3: return launch_fwd<3, 0, 1>(a, s);
            case 5: return launch_fwd<5, 0, 1>(a, s);
        }
        return NSVD_EUNSUPPORTED;
    }
    if (prob.eps <= 0.f) {  // exact Laplacian: D + 2 jet streams
        switch (d.D) {
            case 1: return launch_fwd<3, 1>(a, s);
            case 2: return launch_fwd<4, 1>(a, s);
            case 3: return launch_fwd<5, 1>(a, s);
        }
        return NSVD_EUNSUPPORTED;
    }
    if (d.D == 2 && fwd_kslices(d, B) == 4) {  // configs[0]: 4 x 64 workgroups for layer 0, then 64 for the rest
        a.ks = 4;
        a.kpart = w.kpart;
        {
            const char* e = getenv("NSVD_KSPLIT_FOLD");  // "0": the separate epilogue launch (A/B measurements, tests)
            a.tickets = (e && e[0] == '0') ? nullptr : w.tickets;
        }
        rc = launch_fwd<5, 0, 0, 0, 1>(a, s, 1);  // (the in-run bracket of bench.py spans both launches)
        if (rc) return rc;
        // the rest of the network in SPLIT form - 2 x 64 workgroups of three tiles read the plain-form partials - then
        // the finite-difference epilogue kernel: 128 CUs busy for the latency-bound hidden layers instead of 64
        a.split = d.D;
        a.base_raw = w.base_raw;
        a.kplain = 1;
        rc = launch_fwd<3, 0, 0, 0, 2>(a, s, 2);
        if (rc || a.tickets) return rc;  // (tickets: the last direction group of every tile has formed f, Tf)
        return nsvd_fd_epilogue(w.base_raw, R, x, d.has_exp_mask ? p.scales : nullptr, prob, B, d.D, d.L, f, Tf,
                                save ? w.jac : nullptr, (save && d.has_exp_mask) ? w.dsc : nullptr, s, 1);
    }
    // split-stencil form: D = 3 (7 stencil columns do not fit one workgroup's LDS image), and D = 2 when the plain grid
    // would leave at least half of the CUs without a workgroup (cfg1: 64 -> 128 workgroups of three column tiles)
    if (d.D == 3 || (d.D == 2 && (B / BS) * d.L <= 128)) {
        a.split = d.D;
        a.base_raw = w.base_raw;
        a.ks = d.D == 2 ? fwd_kslices(d, B) : 1;
        if (a.ks == 2) {  // 2 x 128 workgroups for layer 0, then 128 for the rest of the network
            a.kpart = w.kpart;
            {
                const char* e = getenv("NSVD_KSPLIT_FOLD");
                a.tickets = (e && e[0] == '0') ? nullptr : w.tickets;
            }
            rc = launch_fwd<3, 0, 0, 0, 1>(a, s, 1);  // (the in-run bracket of bench.py spans both launches)
            if (rc) return rc;
            rc = launch_fwd<3, 0, 0, 0, 2>(a, s, 2);
            if (rc || a.tickets) return rc;
        } else {
            rc = launch_fwd<3>(a, s);
        }
        if (rc) return rc;
        return nsvd_fd_epilogue(w.base_raw, R, x, d.has_exp_mask ? p.scales : nullptr, prob, B, d.D, d.L, f, Tf,
                                save ? w.jac : nullptr, (save && d.has_exp_mask) ? w.dsc : nullptr, s, 1);
    }
    switch (E) {
        case 3: return launch_fwd<3>(a, s);
        case 5: return launch_fwd<5>(a, s);
    }
    return NSVD_EUNSUPPORTED;
}

// ---- plain model evaluation on the fused kernels (E = 1): out = c * model(x), any input dimension --------------
bool nsvd_fused_model_supported(const nsvd_model_desc& d, int B) {
    if (d.D < 1 || d.D > 64) return false;
    return fused_shape_ok(d, B);
}

int nsvd_fused_model_forward(const nsvd_model_desc& d, const nsvd_params& p, const float* x, int B, float c,
                             float* out, void* ws, int save, hipStream_t s) {
    const FusedWs w = carve_fused(d, B, ws);
    const int F = 2 * d.m;
    int rc = nsvd_fourier_plain(x, p.fourier_B, w.phi, save ? w.phiTc : nullptr, B, d.D, d.m, s);
    if (rc) return rc;
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.phiT = w.phi;
    a.m = d.m;
    a.nlayers = d.nlayers;
    for (int i = 0; i < d.nlayers; ++i) {
        a.W[i] = p.W[i];
        a.b[i] = p.b[i];
        a.zsave[i] = (save && i < d.nlayers - 1) ? w.zsave[i] : nullptr;
    }
    a.x = x;
    a.scales = d.has_exp_mask ? p.scales : nullptr;
    a.prob.hard_mul_const = c;
    a.plain = 1;
    a.B = B; a.D = d.D; a.L = d.L; a.F = F;
    a.f = out;
    a.jac = save ? w.jac : nullptr;
    a.dsc = (save && d.has_exp_mask) ? w.dsc : nullptr;
    // the model shape of the kernel-operator row on enough tiles: the streaming form (pmlp_plain_fwd.h) - weights in
    // registers, two workgroups per CU
    {
        static const char* e = getenv("NSVD_PLAIN_STREAM");
        const int tpw = (e && e[0] == '0') ? 0 : plain_stream_tpw(d, B);
        if (tpw > 0) {
            PlainFwdArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.phi = w.phi;
            pa.W0 = p.W[0]; pa.b0 = p.b[0]; pa.W1 = p.W[1]; pa.b1 = p.b[1]; pa.Wl = p.W[2]; pa.bl = p.b[2];
            pa.x = x; pa.scales = a.scales; pa.D = d.D; pa.c = c;
            pa.out = out; pa.jac = a.jac; pa.dsc = a.dsc;
            pa.z0 = a.zsave[0]; pa.z1 = a.zsave[1];
            pa.B = B; pa.L = d.L; pa.tpw = tpw;
            static bool attr_set = false;
            if (!attr_set) {
                hipError_t er = hipFuncSetAttribute((const void*)pmlp_plain_stream_fwd_kernel,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)PF_LDS_BYTES);
                if (er != hipSuccess) return -(int)er;
                attr_set = true;
            }
            nsvd_prof_begin(s);
            hipLaunchKernelGGL(pmlp_plain_stream_fwd_kernel, dim3((B / BS / tpw) * d.L), dim3(256), PF_LDS_BYTES, s, pa);
            nsvd_prof_end(s);
            NSVD_CHECK_LAUNCH();
            return 0;
        }
    }
    // four sample tiles per workgroup (a head's W_0 / W_i tiles fetched once per 128 samples, the fixed costs of a
    // workgroup - first-chunk latency, weight DMA, epilogue - paid once per 128) whenever that still leaves every CU
    // two workgroups; one tile otherwise
    if (B % (4 * BS) == 0 && (B / (4 * BS)) * d.L >= 512) {
        a.xcd_remap = pick_xcd_remap(d.L, B / (4 * BS), F);
        return launch_fwd<4, 0, 0, 1>(a, s);
    }
    a.xcd_remap = pick_xcd_remap(d.L, B / BS, F);
    return launch_fwd<1, 0, 0, 1>(a, s);
}

#ifdef NSVD_EO_COUNT
extern "C" void nsvd_debug_eo_count(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_eo_count), 8 * sizeof(unsigned long long));
    if (reset) {
        unsigned long long z[8] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_eo_count), z, sizeof(z));
    }
}
#endif
