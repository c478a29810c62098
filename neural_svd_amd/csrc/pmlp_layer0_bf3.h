// Layer 0 of the fused forward on the bf16 MFMA (NSVD_PATH_FUSED_BF16X3). Included by pmlp_fwd.hip
// inside its unnamed namespace, after FwdArgs.
// ================================================================================================
// Layer 0 on the bf16 MFMA with float32-equivalent accuracy (opt-in path NSVD_PATH_FUSED_BF16X3).
// Every float32 operand is split into three bf16 planes by round-to-nearest residuals,
//     x = hi + mid + lo + O(2^-26 |x|),   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (v_cvt_pk_bf16_f32; the residuals are exact in float32), and a product is the six partial products
//     hi hi + hi mid + mid hi + hi lo + lo hi + mid mid          (dropped: mid lo + lo mid + lo lo <= 2^-25 |a b|)
// accumulated in float32 by v_mfma_f32_32x32x16_bf16, smallest terms first. Each partial product of two 8-bit
// significands is exact in the MFMA, so the only errors are the dropped terms and the float32 accumulation - no
// larger than the native fp32 MFMA's own rounding (DESIGN.md 3.5 for the measured parity), at 16/6 of its rate.
//
// Round 4 form. What bounded the round-1 loop (3200 cycles per 32-wide chunk against 1920 of MFMA issue) was the refill
// of its double-buffered LDS stage: 54 KB per chunk through registers and ds_write with one chunk of lead, less than
// the loaded memory latency. Two observations remove most of it:
//   * a wave multiplies ITS OWN 32 rows of W_0 only - the A fragments are shared with no other wave - so the W planes
//     need no LDS at all: w0_split_kernel writes them FRAGMENT-MAJOR ((L, F/32 chunks, 4 waves, 3 planes, 2 k-steps,
//     64 lanes) x 16 B: the six 1 KB fragments of a wave and chunk are 6 KB contiguous) and the K loop loads them
//     straight into registers, four register sets deep: a fragment is requested 2.5 chunks before its MFMAs;
//   * the centre features (8 KB per pair of chunks) are requested 2.5 chunks ahead into a second register set.
// LDS then holds the sample-column planes only (generated in the MFMA shadow from the centre features by angle
// addition, as before): per chunk 30 KB written + 120 KB of fragment reads instead of 54 + 147 KB, and every global
// request of the loop is an ordinary load the compiler counts exactly (no LDS-DMA, no manual vmcnt).
// LDS image of a chunk's sample columns (32 k): per plane [row][32 bf16 + 16 B pad] = 80-B rows (conflict-free
// ds_read_b128 fragment reads: 20-dword stride), 3 planes x NC rows x 2 buffers.
typedef __bf16 nsvd_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int B3_ROW = 80;  // bytes

__device__ __forceinline__ unsigned nsvd_cvt_pk_bf16(float lo, float hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const b2_t h = __builtin_convertvector(f2_t{lo, hi}, b2_t);
    return __builtin_bit_cast(unsigned, h);
}
// the three planes of 4 consecutive-k floats, 8 bytes each
// (tried: v_dot2c_f32_bf16 with (-1, 0) / (0, -1) for "x - element of the packed pair" in one instruction instead of
// shift / mask + subtract: 117 fewer VALU instructions per four chunks, no faster - and the compiler folds the packed
// constant into an inline -1.0 that the instruction applies to both halves: wrong results)
__device__ __forceinline__ void nsvd_bf3_split(const float4 x, uint2& p0, uint2& p1, uint2& p2) {
    p0.x = nsvd_cvt_pk_bf16(x.x, x.y);
    p0.y = nsvd_cvt_pk_bf16(x.z, x.w);
    float4 r;
    r.x = x.x - __uint_as_float(p0.x << 16);
    r.y = x.y - __uint_as_float(p0.x & 0xffff0000u);
    r.z = x.z - __uint_as_float(p0.y << 16);
    r.w = x.w - __uint_as_float(p0.y & 0xffff0000u);
    p1.x = nsvd_cvt_pk_bf16(r.x, r.y);
    p1.y = nsvd_cvt_pk_bf16(r.z, r.w);
    r.x -= __uint_as_float(p1.x << 16);
    r.y -= __uint_as_float(p1.x & 0xffff0000u);
    r.z -= __uint_as_float(p1.y << 16);
    r.w -= __uint_as_float(p1.y & 0xffff0000u);
    p2.x = nsvd_cvt_pk_bf16(r.x, r.y);
    p2.y = nsvd_cvt_pk_bf16(r.z, r.w);
}

// the E rows (stencil points, or jet streams) of one chunk half from the centre pair (sin s, cos c)
template <int E, int JET, int HALF>
__device__ __forceinline__ void nsvd_rows_from_centre(const float4 rs, const float4 rc, const float4 (&cd)[3],
                                                      const float4 (&sd)[3], float4 (&rb)[E]) {
    constexpr int DD = JET ? E - 2 : (E - 1) / 2;
    const float4 u = HALF ? rc : rs, v = HALF ? rs : rc;  // this half's feature and its partner
    rb[0] = u;
    if (JET) {
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            const float sg = HALF ? -1.f : 1.f;  // d sin = B cos, d cos = -B sin
            rb[1 + d] = make_float4(sg * (v.x * cd[d].x), sg * (v.y * cd[d].y), sg * (v.z * cd[d].z), sg * (v.w * cd[d].w));
        }
        rb[E - 1] = make_float4(-(u.x * sd[0].x), -(u.y * sd[0].y), -(u.z * sd[0].z), -(u.w * sd[0].w));
    } else {
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            // sin: x + eps e_d -> s cd + c sd, x - eps e_d -> s cd - c sd; cos: c cd - s sd, c cd + s sd
            const float4 pl = make_float4(fmaf(u.x, cd[d].x, v.x * sd[d].x), fmaf(u.y, cd[d].y, v.y * sd[d].y),
                                          fmaf(u.z, cd[d].z, v.z * sd[d].z), fmaf(u.w, cd[d].w, v.w * sd[d].w));
            const float4 mi = make_float4(fmaf(u.x, cd[d].x, -(v.x * sd[d].x)), fmaf(u.y, cd[d].y, -(v.y * sd[d].y)),
                                          fmaf(u.z, cd[d].z, -(v.z * sd[d].z)), fmaf(u.w, cd[d].w, -(v.w * sd[d].w)));
            rb[1 + 2 * d] = HALF ? mi : pl;
            rb[2 + 2 * d] = HALF ? pl : mi;
        }
    }
}

// Stencil mode only: the 2 DD shifted rows in EVEN / ODD form. With d = eps B_dj,
//     phi(x +- eps e_d) - phi(x):   sin: s (cos d - 1) +- c sin d,   cos: c (cos d - 1) -+ s sin d,
// i.e. an even part u (cos d - 1) (~ d^2 / 2: 2^-14 of the centre at configs[1]) and an odd part +- v sin d (~ d: 2^-7);
// `cm` = cos d - 1 comes from the table as -2 sin^2(d / 2), without the cancellation. Layer 0 is linear, so
//     W phi(x +- eps e_d) = W phi(x) + W even_d +- W odd_d :
// column tile 0 is the centre product (all six partial products), tiles 1 + 2 d / 2 + 2 d the even / odd products of
// direction d - small, so two bf16 planes and three partial products (hi hi + hi mid + mid hi; what is dropped is <=
// 2^-17 of the tile's OWN size) carry them: 12 + 4 x 6 = 36 MFMAs per chunk and wave instead of 60, 11 fragment reads
// per k-step instead of 15. And the tiles stay apart through the whole network (pmlp_fwd.hip: the softplus acts on
// (centre, even, odd) by its Taylor expansion around the centre), so that the quantity the finite-difference stencil
// is after - sum_d [f(x + eps e_d) + f(x - eps e_d) - 2 f(x)] = 2 sum_d even_d - is never formed as the difference of
// rounded large numbers: the float32 stencil noise (a per-point error of ~|f| at eps = 0.01, for the reference's own
// arithmetic too) is gone, the Laplacian stream is accurate to ~1e-5 of itself.
template <int E, int HALF>
__device__ __forceinline__ void nsvd_rows_delta(const float4 rs, const float4 rc, const float4 (&cm)[3],
                                                const float4 (&sd)[3], float4 (&rb)[E]) {
    constexpr int DD = (E - 1) / 2;
    const float4 u = HALF ? rc : rs, v = HALF ? rs : rc;  // this half's feature and its partner
    const float sg = HALF ? -1.f : 1.f;                   // d sin = +cos, d cos = -sin
    rb[0] = u;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
        rb[1 + 2 * d] = make_float4(u.x * cm[d].x, u.y * cm[d].y, u.z * cm[d].z, u.w * cm[d].w);
        rb[2 + 2 * d] = make_float4(sg * (v.x * sd[d].x), sg * (v.y * sd[d].y), sg * (v.z * sd[d].z), sg * (v.w * sd[d].w));
    }
}
// two planes of 4 consecutive-k floats (the perturbation rows)
__device__ __forceinline__ void nsvd_bf2_split(const float4 x, uint2& p0, uint2& p1) {
    p0.x = nsvd_cvt_pk_bf16(x.x, x.y);
    p0.y = nsvd_cvt_pk_bf16(x.z, x.w);
    float4 r;
    r.x = x.x - __uint_as_float(p0.x << 16);
    r.y = x.y - __uint_as_float(p0.x & 0xffff0000u);
    r.z = x.z - __uint_as_float(p0.y << 16);
    r.w = x.w - __uint_as_float(p0.y & 0xffff0000u);
    p1.x = nsvd_cvt_pk_bf16(r.x, r.y);
    p1.y = nsvd_cvt_pk_bf16(r.z, r.w);
}

// W_0 (L, 128, F) float32 -> three bf16 planes, FRAGMENT-MAJOR: for head l, chunk c (the order the K loop visits them:
// pair j of 32-wide sin / cos chunks, k = 32 j .. and m + 32 j ..), wave w (hidden rows 32 w ..), plane p, k-step ks,
// lane (li, hi): the 16 bytes the lane feeds v_mfma_f32_32x32x16_bf16 as its A fragment,
//     P[((((l nch + c) 4 + w) 3 + p) 2 + ks) 64 + lane] = plane_p(W_0[l][32 w + li][k0(c) + 16 ks + 8 hi .. + 7]).
// The hidden layers' weights W_1 .. W_{nh-1} (L, 128, 128) go the same way in the same launch (Wh[j], Ph):
//     Ph[(((((j L + l) 4 + w) 8 + ks) 3 + p) 64 + lane] = plane_p(W_{j+1}[l][32 w + li][16 ks + 8 hi .. + 7]).
// Once per forward call (the weights change every step): the 16 workgroups of a head would otherwise each redo the
// conversion (a wave cannot hide its own VALU work under its own MFMAs: measured 2.5-3 of 4 cycles exposed).
struct W0SplitArgs {
    const float4* W;   // W_0
    uint4* P;
    const float4* Wh[NSVD_MAX_LAYERS];  // W_1 .. (nhid of them)
    uint4* Ph;
    int L, m, nhid;
};
__global__ void __launch_bounds__(256) w0_split_kernel(W0SplitArgs a) {
    const float4* __restrict__ W = a.W;
    uint4* __restrict__ P = a.P;
    const int L = a.L, m = a.m;
    const int F = 2 * m, nch = F / BK;
    const size_t n = (size_t)L * nch * 4 * 2 * 64;  // (l, c, w, ks, lane): one thread item = 8 consecutive k of one row
    // hidden layers: (j, l, w, ks, lane) items
    const size_t nhid_items = (size_t)a.nhid * L * 4 * 8 * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nhid_items; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), ks = (int)((i >> 6) & 7), w = (int)((i >> 9) & 3);
        const size_t jl = i >> 11;  // j * L + l
        const int l = (int)(jl % L), j = (int)(jl / L);
        const int li = lane & 31, hi = lane >> 5;
        const float4* src = a.Wh[j] + (((size_t)l * HID + 32 * w + li) * HID + 16 * ks + 8 * hi) / 4;
        uint2 a0, a1, a2, b0, b1, b2;
        nsvd_bf3_split(src[0], a0, a1, a2);
        nsvd_bf3_split(src[1], b0, b1, b2);
        uint4* dst = a.Ph + (((jl * 4 + w) * 8 + ks) * 3) * 64 + lane;
        dst[0] = make_uint4(a0.x, a0.y, b0.x, b0.y);
        dst[64] = make_uint4(a1.x, a1.y, b1.x, b1.y);
        dst[128] = make_uint4(a2.x, a2.y, b2.x, b2.y);
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), ks = (int)((i >> 6) & 1), w = (int)((i >> 7) & 3);
        const size_t lc = i >> 9;  // l * nch + c
        const int c = (int)(lc % nch), l = (int)(lc / nch);
        const int li = lane & 31, hi = lane >> 5;
        const int k0 = ((c & 1) ? m : 0) + (c >> 1) * BK + 16 * ks + 8 * hi;
        const float4* src = W + (((size_t)l * HID + 32 * w + li) * F + k0) / 4;
        uint2 a0, a1, a2, b0, b1, b2;
        nsvd_bf3_split(src[0], a0, a1, a2);
        nsvd_bf3_split(src[1], b0, b1, b2);
        uint4* dst = P + ((lc * 4 + w) * 3 * 2 + ks) * 64 + lane;
        dst[0] = make_uint4(a0.x, a0.y, b0.x, b0.y);
        dst[2 * 64] = make_uint4(a1.x, a1.y, b1.x, b1.y);
        dst[4 * 64] = make_uint4(a2.x, a2.y, b2.x, b2.y);
    }
}

// No code: ties a value to this point of the instruction stream as ONE 128-bit register tuple. Computations that depend
// only on registers otherwise float freely - hipcc hoists the next step's conversion arithmetic above the barrier to
// right behind the loads that feed it, and then waits there for them. And the tuple matters: with four scalar
// constraints the allocator scatters the components of a 16-byte load's destination, loads into a temporary tuple
// and copies - and the copies wait for the load right behind its issue (seen: vmcnt(0) at the loop's back edge).
typedef float nsvd_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nsvd_pin(nsvd_f32x4& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float4 nsvd_f4(const nsvd_f32x4 v) { return __builtin_bit_cast(float4, v); }

// diagnostic builds only (scripts/dev/build_stamps.sh EXTRA=-DNSVD_BF3_EXP=mask): leave parts of the K loop out to
// price them - 1: the generation of the next pair's sample planes (VALU + LDS stores), 2: the global requests,
// 4: the barrier, 8: the sample fragment reads. Results are then wrong; only the stamps are read.
// Measured on the one-barrier-per-chunk, six-products-everywhere form (cycles per 32-wide chunk, cfg2, 1920 of them
// MFMA issue): everything in 2740; without 1: 2290; without 2: 2500; without 4: 2460; without 8: 2650; MFMAs alone: 1920.
#ifndef NSVD_BF3_EXP
#define NSVD_BF3_EXP 0
#endif
// The K loop walks PAIRS of chunks - the 32 sin features k in [32 q, 32 q + 32) and their 32 cos partners - with ONE
// barrier per pair.
// Column tiles: tile 0 = the centre rows (three planes, six partial products); tiles 1 .. E-1 = in stencil mode the
// even / odd perturbation rows (two planes, three partial products: nsvd_rows_delta above), in jet mode the
// derivative streams (three planes, six products, as tile 0).
// LDS: the sample-column planes of a pair, [buffer 2][half 2][plane][rows][64 B], unpadded: the 16-byte slot of row r
// holds k-octet slot ^ ((r >> 2) & 3) (applied on the 8-byte stores and on the fragment reads), which makes the
// ds_read_b128 of 16 consecutive rows hit 16 distinct 4-bank groups. Plane 0 and 1: NC rows, plane 2: the tiles that
// have one (32 rows in stencil mode, NC in jet mode).
template <int E, int JET>
__device__ __forceinline__ void nsvd_layer0_bf3(const FwdArgs& a, f32x16 (&acc)[E], char* lds, int l, int b0) {
    constexpr int NC = E * BS;
    constexpr bool DELTA = !JET;               // stencil columns as centre + perturbation
    constexpr int DD = JET ? E - 2 : (E - 1) / 2;
    constexpr int P2ROWS = DELTA ? BS : NC;    // rows of plane 2
    constexpr int PLANE = NC * 64, HALFB = 2 * PLANE + P2ROWS * 64, B_BUF = 2 * HALFB;  // bytes
    char* Bs = lds;
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s_row = tid >> 3, s_c4 = tid & 7;
    // every request of the loop: uniform base + one 32-bit byte offset per thread, fixed for the whole loop
    const char* b_u = reinterpret_cast<const char*>(a.phiT + (size_t)b0 * a.F);
    const char* t_u = reinterpret_cast<const char*>(a.sctab);
    const unsigned offB = (unsigned)(s_row * a.F + 4 * s_c4) * 4u, offT = (unsigned)(4 * s_c4) * 4u;
    const int mm = a.m, nch = a.F / BK, npair = nch / 2;
    // this wave's A fragments: 6 x 1 KB per chunk, contiguous; chunk stride 4 waves x 6 KB
    const char* wf_u = reinterpret_cast<const char*>(a.w0p) + ((size_t)l * nch * 4 + w) * (6 * 64 * 16);
    const unsigned offW = (unsigned)lane * 16u;
    constexpr size_t WCH = 4 * 6 * 64 * 16;  // bytes per chunk
    uint4 fa[4][6];                           // [register set = chunk & 3][2 p + ks]
    // centre features + stencil constants of a pair, set = pair & 1; cd: cos(eps B) in jet mode... the table's first
    // slot: B_dj (jets), cos(eps B_dj) - 1 (stencil, DELTA)
    nsvd_f32x4 rs[2], rc[2], cd[2][3], sd[2][3];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        rs[s] = rc[s] = nsvd_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < 3; ++d) cd[s][d] = sd[s][d] = nsvd_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_w = [&](auto set, int c) {  // fragments of chunk c (clamped: the tail requests an in-range chunk again)
        constexpr int S = decltype(set)::value;
        const char* p = wf_u + (size_t)(c < nch ? c : nch - 1) * WCH;
#pragma unroll
        for (int i = 0; i < 6; ++i) fa[S][i] = *reinterpret_cast<const uint4*>(p + i * 1024 + offW);
    };
    auto load_f = [&](auto set, int pair) {  // centre features and stencil constants of pair `pair`
        constexpr int S = decltype(set)::value;
        const int kp = (pair < npair ? pair : npair - 1) * BK;
        rs[S] = *reinterpret_cast<const nsvd_f32x4*>(b_u + (size_t)kp * 4 + offB);
        rc[S] = *reinterpret_cast<const nsvd_f32x4*>(b_u + (size_t)(mm + kp) * 4 + offB);
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            // stencil mode: cos - 1 from the table's third block (D, m) behind the (D, 2, m) cos / sin pairs
            const size_t c_at = DELTA ? (size_t)((2 * DD + d) * mm + kp) : (size_t)((2 * d) * mm + kp);
            cd[S][d] = *reinterpret_cast<const nsvd_f32x4*>(t_u + c_at * 4 + offT);
            sd[S][d] = *reinterpret_cast<const nsvd_f32x4*>(t_u + (size_t)((2 * d + 1) * mm + kp) * 4 + offT);
        }
    };
    // sample-column fragments of one 16-wide k step (k-step s = 2 half + ks of a pair), two register sets; plane 2 of
    // the perturbation tiles does not exist (never read)
    nsvd_bf16x8 fb[2][E][3];
    const int rsw = (li >> 2) & 3;  // the swizzle of this lane's rows (32 e does not touch bits 2..3)
    auto frags = [&](auto set, int buf, int s) {
        constexpr int S = decltype(set)::value;
        const char* Bp = Bs + buf * B_BUF + (s >> 1) * HALFB + li * 64 + ((((2 * (s & 1) + hi) ^ rsw)) << 4);
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                if (p < 2 || !DELTA || e == 0)
                    fb[S][e][p] = *reinterpret_cast<const nsvd_bf16x8*>(Bp + p * PLANE + e * (32 * 64));
    };
    constexpr int NFR = DELTA ? 3 + 2 * (E - 1) : 3 * E;  // fragment reads per k-step
    // store address of this thread's 4 k of row s_row (+ 32 e): slot (s_c4 >> 1) swizzled by the row, 8-byte half
    const int wofs = s_row * 64 + ((((s_c4 >> 1) ^ ((s_row >> 2) & 3))) << 4) + 8 * (s_c4 & 1);
    auto put_row = [&](char* Bh, int e, const float4 v) {  // Bh: the (buffer, half) image
        char* q = Bh + wofs + e * (32 * 64);
        if (DELTA && e > 0) {
            uint2 p0, p1;
            nsvd_bf2_split(v, p0, p1);
            *reinterpret_cast<uint2*>(q) = p0;
            *reinterpret_cast<uint2*>(q + PLANE) = p1;
        } else {
            uint2 p0, p1, p2;
            nsvd_bf3_split(v, p0, p1, p2);
            *reinterpret_cast<uint2*>(q) = p0;
            *reinterpret_cast<uint2*>(q + PLANE) = p1;
            *reinterpret_cast<uint2*>(q + 2 * PLANE) = p2;
        }
    };
    auto gen_rows = [&](auto half, int FSi, float4 (&rb)[E]) {
        constexpr int HALF = decltype(half)::value;
        const float4 cdf[3] = {nsvd_f4(cd[FSi][0]), nsvd_f4(cd[FSi][1]), nsvd_f4(cd[FSi][2])};
        const float4 sdf[3] = {nsvd_f4(sd[FSi][0]), nsvd_f4(sd[FSi][1]), nsvd_f4(sd[FSi][2])};
        if constexpr (DELTA) nsvd_rows_delta<E, HALF>(nsvd_f4(rs[FSi]), nsvd_f4(rc[FSi]), cdf, sdf, rb);
        else nsvd_rows_from_centre<E, JET, HALF>(nsvd_f4(rs[FSi]), nsvd_f4(rc[FSi]), cdf, sdf, rb);
    };
    // stencil mode: the centre tile's six products alternate between TWO accumulators (consecutive MFMAs on one
    // accumulator wait for each other); tiles 1 .. E-1 start from zero and receive the centre product after the loop
    f32x16 accb;
    if constexpr (DELTA) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accb[r] = 0.f;
#pragma unroll
            for (int e = 1; e < E; ++e) acc[e][r] = 0.f;
        }
    }
    // MFMA groups of one k-step. Jets: six groups of E (one partial product each, smallest first). Stencil: four groups -
    // {centre lo hi, perturbations mid hi'} {centre hi lo, perturbations hi mid'} {centre mid mid, perturbations hi hi'}
    // {centre mid hi, hi mid, hi hi} - 18 MFMAs.
    constexpr int NG = DELTA ? 4 : 6;          // groups per k-step
    constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};  // (A plane, B plane), smallest first
    constexpr int DA[3] = {1, 0, 0}, DB[3] = {0, 1, 0};                    // the perturbation tiles' three products
    auto mma_group = [&](auto set, auto fset, int ks, int gk) {  // group gk of k-step (fragment set fset, A set `set`)
        constexpr int S = decltype(set)::value;
        constexpr int FB = decltype(fset)::value;
        auto A = [&](int p) { return __builtin_bit_cast(nsvd_bf16x8, fa[S][2 * p + ks]); };
        if constexpr (DELTA) {
            if (gk < 3) {
                if (gk & 1) accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[gk]), fb[FB][0][TB[gk]], accb, 0, 0, 0);
                else acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[gk]), fb[FB][0][TB[gk]], acc[0], 0, 0, 0);
#pragma unroll
                for (int e = 1; e < E; ++e)
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(DA[gk]), fb[FB][e][DB[gk]], acc[e], 0, 0, 0);
            } else {
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[3]), fb[FB][0][TB[3]], accb, 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[4]), fb[FB][0][TB[4]], acc[0], 0, 0, 0);
                accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[5]), fb[FB][0][TB[5]], accb, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; ++e)
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A(TA[gk]), fb[FB][e][TB[gk]], acc[e], 0, 0, 0);
        }
    };
    // One PAIR q (chunks 2 q, 2 q + 1), PQ = q & 1 named at compile time: 4 k-steps (2 halves x 2) of NG MFMA groups
    // on fragment sets 2 PQ, 2 PQ + 1 and sample buffer PQ, and in their shadow (g = global group index, 0 .. 4 NG - 1)
    //   groups 0 .. 2E-1:      sample row g of pair q + 1 (half g / E, row g % E) into the other buffer;
    //   group 2E:              the request for the centre features of pair q + 3 (their set was just consumed);
    //   first group of k-step s (s = 0, 1, 2): the sample fragments of k-step s + 1 of this pair;
    //   first group of k-step 2, last group: the requests for the A fragments of chunks 2 q + 4, 2 q + 5 (their sets
    //                          just went idle) - three chunks before their first use;
    //   first group of k-step 3: the pair's only barrier (every wave has read this buffer and written the next), then
    //                          the k-step-0 sample fragments of pair q + 1.
    auto pstep = [&](auto pq, auto do_store, int q) {
        constexpr int PQ = decltype(pq)::value;
        constexpr bool ST = decltype(do_store)::value;
        constexpr int FS = PQ ^ 1;                 // feature set of pair q + 1
        char* Bn = Bs + (PQ ^ 1) * B_BUF;          // where pair q + 1 is written
        f32x16& accb_l = accb;                     // (named here: an asm operand alone does not capture it)
        float4 rb[E];
        if (ST) {
            nsvd_pin(rs[FS]);
            nsvd_pin(rc[FS]);
#pragma unroll
            for (int d = 0; d < DD; ++d) {
                nsvd_pin(cd[FS][d]);
                nsvd_pin(sd[FS][d]);
            }
        }
        static_assert(2 * E + 1 <= 4 * NG - 1, "the pair's groups must hold the generation of the next pair");
#pragma unroll
        for (int g = 0; g < 4 * NG; ++g) {
            const int s = g / NG, gk = g % NG, h = s >> 1, ks = s & 1;
            const bool rd = !(NSVD_BF3_EXP & 8);
            if (s == 3 && gk == 0 && !(NSVD_BF3_EXP & 4)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // fragment set: k-step s reads set s & 1; A fragments: register set 2 PQ + h
            if (s & 1) {
                if (h) mma_group(std::integral_constant<int, 2 * PQ + 1>{}, std::integral_constant<int, 1>{}, ks, gk);
                else mma_group(std::integral_constant<int, 2 * PQ>{}, std::integral_constant<int, 1>{}, ks, gk);
            } else {
                if (h) mma_group(std::integral_constant<int, 2 * PQ + 1>{}, std::integral_constant<int, 0>{}, ks, gk);
                else mma_group(std::integral_constant<int, 2 * PQ>{}, std::integral_constant<int, 0>{}, ks, gk);
            }
            bool did_frags = false;
            if (gk == 0 && rd) {
                if (s == 0) { frags(std::integral_constant<int, 1>{}, PQ, 1); did_frags = true; }
                if (s == 1) { frags(std::integral_constant<int, 0>{}, PQ, 2); did_frags = true; }
                if (s == 2) { frags(std::integral_constant<int, 1>{}, PQ, 3); did_frags = true; }
                if (s == 3 && ST) { frags(std::integral_constant<int, 0>{}, PQ ^ 1, 0); did_frags = true; }
            }
            const bool gen = ST && !(NSVD_BF3_EXP & 1) && g < 2 * E;
            if (gen) {
                if (g == 0) gen_rows(std::integral_constant<int, 0>{}, FS, rb);
                if (g == E) gen_rows(std::integral_constant<int, 1>{}, FS, rb);
                put_row(Bn + (g / E) * HALFB, g % E, rb[g % E]);
            }
            bool did_loads = false;
            if (!(NSVD_BF3_EXP & 2)) {
                // (the feature set of pair q + 1 is free once its rows are generated: pair q + 3 goes there)
                if (g == 2 * E) { load_f(std::integral_constant<int, FS>{}, q + 3); did_loads = true; }
                if (s == 2 && gk == 1) { load_w(std::integral_constant<int, 2 * PQ>{}, 2 * q + 4); did_loads = true; }
                if (g == 4 * NG - 1) { load_w(std::integral_constant<int, 2 * PQ + 1>{}, 2 * q + 5); did_loads = true; }
            }
            // inside the group: every MFMA followed by its share of the group's other work (a wave issues in order:
            // VALU placed behind all the MFMAs would start only when the last one has issued)
            const int nm = DELTA ? (gk < 3 ? E : 3) : E;  // MFMAs of this group
#pragma unroll
            for (int e = 0; e < nm; ++e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (did_frags) {
                    if (nm == E) __builtin_amdgcn_sched_group_barrier(0x100, (NFR + E - 1) / E, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, (NFR + 2) / 3, 0);
                }
                if (gen) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                if (did_loads) {
                    if (nm == E) __builtin_amdgcn_sched_group_barrier(0x020, (6 + E - 1) / E, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // keep the groups apart (no code: the accumulators are tied to one statement, so that the chains advance
            // together; left alone the compiler runs dependent MFMAs of one chain back to back)
            if constexpr (E == 3) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]));
            if constexpr (E == 4) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
            if constexpr (E == 5)
                asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]));
            if constexpr (DELTA) asm volatile("" : "+a"(accb_l));
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    using S3 = std::integral_constant<int, 3>;
    using T1 = std::integral_constant<bool, true>;
    using T0 = std::integral_constant<bool, false>;
    // prologue: fragments of chunks 0..3, features of pairs 0, 1 (and 2, once pair 0's are consumed), planes of pair 0
    load_f(P0{}, 0);
    load_w(P0{}, 0);
    load_w(P1{}, 1);
    load_f(P1{}, 1);
    load_w(S2{}, 2);
    load_w(S3{}, 3);
    {
        float4 rb[E];
        gen_rows(std::integral_constant<int, 0>{}, 0, rb);
#pragma unroll
        for (int e = 0; e < E; ++e) put_row(Bs, e, rb[e]);
        gen_rows(std::integral_constant<int, 1>{}, 0, rb);
#pragma unroll
        for (int e = 0; e < E; ++e) put_row(Bs + HALFB, e, rb[e]);
    }
    load_f(P0{}, 2);
    __syncthreads();
    frags(P0{}, 0, 0);
    NSVD_STAMP(1)
    int q = 0;
    for (; q + 2 < npair; q += 2) {  // npair is even (F a multiple of 128); branch-free steady state
        pstep(P0{}, T1{}, q);
        pstep(P1{}, T1{}, q + 1);
    }
    pstep(P0{}, T1{}, q);
    pstep(P1{}, T0{}, q + 1);
    if constexpr (DELTA) {
        // tile 0: b + W phi(x); tiles 1 + 2 d, 2 + 2 d: the even / odd perturbations of direction d, kept apart
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] += accb[r];
    }
    __syncthreads();
}
