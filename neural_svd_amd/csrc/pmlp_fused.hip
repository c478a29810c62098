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
#include <string.h>
#include <type_traits>
#include "nsvd_kernels.h"
#include "fd_math.h"
#include "evd_math.h"
#include "opt_math.h"
#include "tile_nt.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HID = 128;      // hidden width
constexpr int BS = 32;        // base samples per workgroup
constexpr int BK = 32;        // layer-0 K chunk
constexpr int A_LD = BK + 4;  // padded row of the W_0 tile (floats)

struct FwdArgs {
    const float* phiT;   // (B, F) Fourier features of the CENTRE rows, sample-major: [sin(x.B) | cos(x.B)]
    const float* sctab;  // (D, 2, m): cos(eps B_dj), sin(eps B_dj) - the stencil rows are built from the centre
                         // features by angle addition while the layer-0 tiles are staged
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
    const unsigned short* w0p;  // BF3 only: W_0 pre-split into three bf16 planes, (3, L, 128, F), by w0_split_kernel
    size_t w0_plane;            // elements per plane
    int plain;      // E = 1 instance only: out = hard_mul_const * base * mask (WaveFunctions.forward), no Hamiltonian;
                    // f receives the output, jac / dsc its derivatives w.r.t. base / scales
    int xcd_remap;  // 0: plain mapping; else HX = number of head groups across the 8 XCDs (1, 2, 4 or 8)
    unsigned long long* stamps;  // diagnostic build only (NSVD_FWD_STAMPS): per-workgroup s_memtime stamps
};

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

// ================================================================================================
// Layer 0 on the bf16 MFMA with float32-equivalent accuracy (opt-in path NSVD_PATH_FUSED_BF16X3).
// Every float32 operand is split into three bf16 planes by round-to-nearest residuals,
//     x = hi + mid + lo + O(2^-26 |x|),   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (v_cvt_pk_bf16_f32; the residuals are exact in float32), and a product is the six partial products
//     hi hi + hi mid + mid hi + hi lo + lo hi + mid mid          (dropped: mid lo + lo mid + lo lo <= 2^-25 |a b|)
// accumulated in float32 by v_mfma_f32_32x32x16_bf16, smallest terms first. Each partial product of two 8-bit
// significands is exact in the MFMA, so the only errors are the dropped terms and the float32 accumulation - no
// larger than the native fp32 MFMA's own rounding (DESIGN.md 3.7 for the measured parity), at 16/6 of its rate.
// LDS image of a chunk (32 k): per plane [row][32 bf16 + 16 B pad] = 80-B rows (conflict-free ds_read_b128 fragment
// reads: 20-dword stride), planes x buffers: W tile 2 x 3 x 128 rows, sample columns 2 x 3 x NC rows.
typedef __bf16 nsvd_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int B3_ROW = 80;  // bytes

__device__ __forceinline__ unsigned nsvd_cvt_pk_bf16(float lo, float hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const b2_t h = __builtin_convertvector(f2_t{lo, hi}, b2_t);
    return __builtin_bit_cast(unsigned, h);
}
// the three planes of 4 consecutive-k floats, 8 bytes each
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

// W_0 (L, 128, F) float32 -> three bf16 planes (3, L, 128, F): once per forward call (the weights change every step),
// so that the forward's workgroups - 16 per head, all streaming the same W_0 - load ready-made planes instead of each
// converting them again (a wave cannot hide its own VALU work under its own MFMAs: measured 2.5-3 of 4 cycles exposed).
__global__ void __launch_bounds__(256) w0_split_kernel(const float4* __restrict__ W, uint2* __restrict__ P, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        uint2 p0, p1, p2;
        nsvd_bf3_split(W[i], p0, p1, p2);
        P[i] = p0;
        P[n4 + i] = p1;
        P[2 * n4 + i] = p2;
    }
}

// No code: ties a register quad to this point of the instruction stream. Computations that depend only on registers
// otherwise float freely - hipcc hoists the next step's conversion arithmetic above the barrier to right behind the
// loads that feed it, and then waits there for them.
__device__ __forceinline__ void nsvd_pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

template <int E, int JET>
__device__ __forceinline__ void nsvd_layer0_bf3(const FwdArgs& a, f32x16 (&acc)[E], char* lds, int l, int b0) {
    constexpr int NC = E * BS;
    constexpr int DD = JET ? E - 2 : (E - 1) / 2;
    constexpr int A_BUF = 3 * HID * B3_ROW, B_BUF = 3 * NC * B3_ROW;  // bytes per buffer
    char* As = lds;                 // [2][3][128][80 B]
    char* Bs = lds + 2 * A_BUF;     // [2][3][NC][80 B]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 31, hi = lane >> 5;
    const int s_row = tid >> 3, s_c4 = tid & 7;
    const float* b_src = a.phiT + (size_t)(b0 + s_row) * a.F + 4 * s_c4;
    const float* t_src = a.sctab + 4 * s_c4;
    const int mm = a.m, nch = a.F / BK;
    float4 rs, rc, cd[3], sd[3];
    uint4 rw0, rw1, rw2, rw3, rw4, rw5;  // W_0 planes of the chunk: 3 planes x 2 pieces of 16 B (8 bf16) per thread
    rw0 = rw1 = rw2 = rw3 = rw4 = rw5 = make_uint4(0u, 0u, 0u, 0u);
    rs = rc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int d = 0; d < 3; ++d) cd[d] = sd[d] = make_float4(0.f, 0.f, 0.f, 0.f);
    // W tile of a chunk, per plane 128 rows x 64 B: piece q (16 B) of row r for thread index r * 4 + q (2 per thread)
    const int w_row = tid >> 2, w_q = tid & 3;
    const unsigned short* w_src = a.w0p + ((size_t)l * HID + w_row) * a.F + 8 * w_q;
    const size_t w_half = (size_t)64 * a.F;  // second piece: row + 64

    auto load = [&](int c, auto half) {  // chunk c = pair (c >> 1), half (c & 1)
        constexpr int HALF = decltype(half)::value;
        const int kp = (c >> 1) * BK;
        const unsigned short* pw = w_src + (HALF ? mm : 0) + kp;
        rw0 = *reinterpret_cast<const uint4*>(pw);
        rw1 = *reinterpret_cast<const uint4*>(pw + w_half);
        rw2 = *reinterpret_cast<const uint4*>(pw + a.w0_plane);
        rw3 = *reinterpret_cast<const uint4*>(pw + a.w0_plane + w_half);
        rw4 = *reinterpret_cast<const uint4*>(pw + 2 * a.w0_plane);
        rw5 = *reinterpret_cast<const uint4*>(pw + 2 * a.w0_plane + w_half);
        if (!HALF) {
            rs = *reinterpret_cast<const float4*>(b_src + kp);
            rc = *reinterpret_cast<const float4*>(b_src + mm + kp);
#pragma unroll
            for (int d = 0; d < DD; ++d) {
                cd[d] = *reinterpret_cast<const float4*>(t_src + (2 * d) * mm + kp);
                sd[d] = *reinterpret_cast<const float4*>(t_src + (2 * d + 1) * mm + kp);
            }
        }
    };
    auto store_w = [&](int buf, int i) {  // piece i (plane i / 2, row half i % 2) of the W tile in registers
        char* Ab = As + buf * A_BUF + ((i >> 1) * HID + w_row + 64 * (i & 1)) * B3_ROW + 16 * w_q;
        *reinterpret_cast<uint4*>(Ab) = i == 0 ? rw0 : i == 1 ? rw1 : i == 2 ? rw2 : i == 3 ? rw3 : i == 4 ? rw4 : rw5;
    };
    auto store = [&](int buf, auto half) {  // split into planes and write the chunk held in registers (prologue)
        constexpr int HALF = decltype(half)::value;
        char* Bb = Bs + buf * B_BUF + s_row * B3_ROW + 8 * s_c4;
#pragma unroll
        for (int i = 0; i < 6; ++i) store_w(buf, i);
        float4 rb[E];
        nsvd_rows_from_centre<E, JET, HALF>(rs, rc, cd, sd, rb);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            uint2 p0, p1, p2;
            nsvd_bf3_split(rb[e], p0, p1, p2);
            *reinterpret_cast<uint2*>(Bb + (0 * NC + 32 * e) * B3_ROW) = p0;
            *reinterpret_cast<uint2*>(Bb + (1 * NC + 32 * e) * B3_ROW) = p1;
            *reinterpret_cast<uint2*>(Bb + (2 * NC + 32 * e) * B3_ROW) = p2;
        }
    };
    // operand fragments of the two 16-wide k steps of a chunk; those of k step 0 are fetched one chunk ahead
    nsvd_bf16x8 fa[2][3], fb[2][E][3];
    auto frags = [&](int ks, int buf) {
        const char* Ap = As + buf * A_BUF + (32 * w + li) * B3_ROW + 16 * hi;
        const char* Bp = Bs + buf * B_BUF + li * B3_ROW + 16 * hi;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            fa[ks][p] = *reinterpret_cast<const nsvd_bf16x8*>(Ap + p * HID * B3_ROW + 32 * ks);
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                fb[ks][e][p] = *reinterpret_cast<const nsvd_bf16x8*>(Bp + (p * NC + 32 * e) * B3_ROW + 32 * ks);
    };
    // One chunk: the 12 groups of E MFMAs (2 k-steps x 6 partial products) of buffer `buf`, and behind them the NEXT
    // chunk (held in registers) on its way into buffer `buf ^ 1`: behind group g < E the stencil row g (generated from
    // the centre features, split into three planes, three 8-byte LDS stores), behind group g < 6 one ready-made 16-byte
    // piece of the W_0 planes; behind group 6 the global loads of the chunk after next, behind group 7 the barrier,
    // behind group 8 the first fragments of the next chunk. The fences pin that placement (left alone hipcc issues
    // all MFMAs first and the conversion instructions after them).
    auto step = [&](int buf, auto half, auto do_store, auto do_load, int cload) {
        constexpr int HALF = decltype(half)::value;      // which half of its pair the NEXT chunk is
        constexpr bool ST = decltype(do_store)::value;
        constexpr bool LD = decltype(do_load)::value;    // fetch chunk cload (the one after next) once the registers are free
        char* Bb = Bs + (buf ^ 1) * B_BUF + s_row * B3_ROW + 8 * s_c4;
        float4 rb[E];
        if (ST) {
            nsvd_pin(rs);
            nsvd_pin(rc);
#pragma unroll
            for (int d = 0; d < DD; ++d) {
                nsvd_pin(cd[d]);
                nsvd_pin(sd[d]);
            }
        }
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};  // (A plane, B plane), smallest first
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int ks = g / 6, t = g % 6;
#pragma unroll
            for (int e = 0; e < E; ++e)
                acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][TA[t]], fb[ks][e][TB[t]], acc[e], 0, 0, 0);
            if (g == 1) frags(1, buf);
            // The chunk's only barrier sits behind group 7: by then every wave has issued all its reads of this buffer
            // (k step 1 was fetched behind group 1) and has written its share of the next chunk (groups 0..5), so the
            // first fragments of the NEXT chunk are fetched here, under the last four groups. (With the barrier at the
            // chunk boundary all four waves start each chunk waiting on the same 18 fragment reads per wave, ~600
            // cycles of LDS bandwidth with the matrix pipe empty.)
            if (g == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (ST && g == 8) frags(0, buf ^ 1);
            if (ST) {
                // behind group g: the sample row g of the next chunk (generated, split, three 8-byte stores) for
                // g < E, then one ready-made 16-byte piece of its W planes per group
                if (g == 0) nsvd_rows_from_centre<E, JET, HALF>(rs, rc, cd, sd, rb);
                if (g < E) {
                    uint2 p0, p1, p2;
                    nsvd_bf3_split(rb[g < E ? g : 0], p0, p1, p2);
                    *reinterpret_cast<uint2*>(Bb + (0 * NC + 32 * g) * B3_ROW) = p0;
                    *reinterpret_cast<uint2*>(Bb + (1 * NC + 32 * g) * B3_ROW) = p1;
                    *reinterpret_cast<uint2*>(Bb + (2 * NC + 32 * g) * B3_ROW) = p2;
                }
                if (g < 6) store_w(buf ^ 1, g);
            }
            // the registers are free from here: the chunk after next is requested half a chunk (~1000 cycles) before
            // the next step starts to convert it (requested at the top of that step, the wave - which issues in order -
            // sits out the whole load latency at its first conversion instruction with the matrix pipe drained)
            if (LD && g == 6) {
                if constexpr (HALF) load(cload, std::integral_constant<int, 0>{});
                else load(cload, std::integral_constant<int, 1>{});
            }
            // inside the group: every MFMA followed by its share of the group's other work (a wave issues in order:
            // VALU placed behind all E MFMAs would start only when the last one has issued)
#pragma unroll
            for (int e = 0; e < E; ++e) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (g == 1 || (ST && g == 8)) __builtin_amdgcn_sched_group_barrier(0x100, (3 + 3 * E + E - 1) / E, 0);
                if (ST && g < E) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // keep the groups apart (no code: the accumulators are tied to one statement, so that the E chains advance
            // together; left alone the compiler runs three dependent MFMAs of one chain back to back at the chunk start)
            if constexpr (E == 3) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]));
            if constexpr (E == 4) asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]));
            if constexpr (E == 5)
                asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]));
        }
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using T1 = std::integral_constant<bool, true>;
    using T0 = std::integral_constant<bool, false>;
    load(0, H0{});
    store(0, H0{});
    __syncthreads();
    load(1, H1{});
    frags(0, 0);
    int c = 0;
    for (; c + 2 < nch; c += 2) {  // nch is even: pairs (sin chunk, cos chunk); branch-free steady state
        step(0, H1{}, T1{}, T1{}, c + 2);  // chunk c; converts chunk c + 1 (registers) into buffer 1, fetches c + 2
        step(1, H0{}, T1{}, T1{}, c + 3);  // chunk c + 1; converts chunk c + 2 into buffer 0, fetches c + 3
    }
    step(0, H1{}, T1{}, T0{}, 0);
    step(1, H0{}, T0{}, T0{}, 0);
    __syncthreads();
}

// BF3 = 1: layer 0 on the bf16 MFMA with three-way split operands (nsvd_layer0_bf3 above); everything after layer 0 is
//   the same code. Its stage buffers are larger, so the W tile of the hidden layers aliases their tail (one extra
//   barrier after the K loop).
template <int E, int JET, int BF3 = 0>
__global__ void __launch_bounds__(256, 1) pmlp_fused_fwd_kernel(FwdArgs a) {
    constexpr int NC = E * BS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][128][A_LD]   W_0 tile, k contiguous
    float* Bs = smem + 2 * HID * A_LD;      // [2][NC][A_LD]    phi tile (rows = sample columns), k contiguous
    float* Hs = smem;                       // [NC][H_LD]       activations, k contiguous (aliases the stage buffers)
    constexpr int STAGE = 2 * HID * A_LD + 2 * NC * A_LD;
    constexpr int HSZ = NC * H_LD;
    constexpr int STAGE3 = (2 * 3 * (HID + NC) * B3_ROW) / 4;  // floats
    constexpr int WL_OFF = BF3 ? HSZ : (STAGE > HSZ ? STAGE : HSZ);
    float* Wl = smem + WL_OFF;   // [4 waves][32][128]  next layer's W rows of each wave (XOR-swizzled chunks)
    constexpr int RED_OFF = BF3 ? (STAGE3 > HSZ + HID * HID ? STAGE3 : HSZ + HID * HID) : WL_OFF + HID * HID;
    float* red = smem + RED_OFF;                      // [4][NC]
    float* outs = red + 4 * NC;                       // [NC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nsb = a.B / BS;
    int l, sb;
    if (a.xcd_remap) {
        // blocks b and b+8 share an XCD (round-robin dispatch: speed only, never correctness). XCD x gets the
        // head group x % HX and the sample-block group x / HX, so its L2 sees L/HX W_0 slabs and nsb/SX phi
        // slabs instead of everything (HX * SX = 8, chosen on the host to minimise bytes per XCD).
        const int HX = a.xcd_remap, SX = 8 / HX;
        const int x = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int hpg = a.L / HX, spg = nsb / SX;  // heads / sample blocks per group
        l = (x % HX) * hpg + slot % hpg;
        sb = (x / HX) * spg + slot / hpg;
    } else {
        l = blockIdx.x / nsb;
        sb = blockIdx.x - l * nsb;
    }
    const int b0 = sb * BS;

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
            for (int e = 0; e < E; ++e) acc[e][r] = (JET && e > 0) ? 0.f : bv;
        }
    }

    if constexpr (BF3) {
        nsvd_layer0_bf3<E, JET>(a, acc, reinterpret_cast<char*>(smem), l, b0);
        __syncthreads();  // the W-tile DMA below lands in the tail of the stage buffers
    } else {
    // ------------------------------------------------------------------ layer 0: K = F in chunks of BK
    // Both operands are k-contiguous rows (W_0[l][n][:] and phi[r][:]): each thread moves one float4 of a
    // 32-row slab per step, global -> registers -> LDS; chunk c+1 is in flight while chunk c is multiplied.
    // Named registers, no arrays: hipcc leaves a conditionally written float4 array in scratch.
    const float* W0 = a.W[0] + (size_t)l * HID * a.F;
    const int nch = a.F / BK;
    // K runs over PAIRS of chunks: 32 sin features k in [32 p, 32 p + 32) and their 32 cos partners m + k. The
    // pair is loaded once (centre row only: 2 float4 per thread + the per-frequency constants) and the E stencil
    // rows of both chunks are generated in registers by angle addition,
    //   sin(t +- d) = sin t cos d +- cos t sin d,  cos(t +- d) = cos t cos d -+ sin t sin d,   d = eps B_dj,
    // with the same float32 expressions the feature kernel used to evaluate: phi(x +- eps e_d) is never stored.
    // (5x less feature traffic per tile; the feature kernel writes B x F instead of E x B x F.)
    constexpr int DD = JET ? E - 2 : (E - 1) / 2;  // input dimensions
    float4 ra0, ra1, ra2, ra3, rs, rc, cd0, sd0, cd1, sd1, cd2, sd2;
    ra0 = ra1 = ra2 = ra3 = rs = rc = cd0 = sd0 = cd1 = sd1 = cd2 = sd2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s_row = tid >> 3, s_c4 = tid & 7;  // 32 rows x 8 float4 per slab
    const float* a_src = W0 + (size_t)s_row * a.F + 4 * s_c4;
    const float* b_src = a.phiT + (size_t)(b0 + s_row) * a.F + 4 * s_c4;  // centre features, (B, F) row-major
    const float* t_src = a.sctab + 4 * s_c4;
    const size_t a_step = (size_t)32 * a.F;
    const int mm = a.m;
#define NSVD_LDG(p) (*reinterpret_cast<const float4*>(p))
// chunk c = pair (c >> 1), half (c & 1): HALF = 0 the sin features, 1 their cos partners
#define NSVD_LOAD_CHUNK(c, HALF)                                                 \
    {                                                                            \
        const int kp_ = ((c) >> 1) * BK;                                         \
        const float* pa_ = a_src + ((HALF) ? mm : 0) + kp_;                      \
        ra0 = NSVD_LDG(pa_);                                                     \
        ra1 = NSVD_LDG(pa_ + a_step);                                            \
        ra2 = NSVD_LDG(pa_ + 2 * a_step);                                        \
        ra3 = NSVD_LDG(pa_ + 3 * a_step);                                        \
        if (!(HALF)) {                                                           \
            rs = NSVD_LDG(b_src + kp_);                                          \
            rc = NSVD_LDG(b_src + mm + kp_);                                     \
            if (DD > 0) cd0 = NSVD_LDG(t_src + kp_);                             \
            if (DD > 0) sd0 = NSVD_LDG(t_src + mm + kp_);                        \
            if (DD > 1) cd1 = NSVD_LDG(t_src + 2 * mm + kp_);                    \
            if (DD > 1) sd1 = NSVD_LDG(t_src + 3 * mm + kp_);                    \
            if (DD > 2) cd2 = NSVD_LDG(t_src + 4 * mm + kp_);                    \
            if (DD > 2) sd2 = NSVD_LDG(t_src + 5 * mm + kp_);                    \
        }                                                                        \
    }
#define NSVD_STS(p, v) (*reinterpret_cast<float4*>(p) = (v))
// u cd + v sd and u cd - v sd, componentwise, as fmaf(u, cd, +-(v * sd)) (the feature kernel's expressions)
#define NSVD_PM(plus, minus, u, v, cd, sd)                                                         \
    {                                                                                              \
        plus = make_float4(fmaf(u.x, cd.x, v.x * sd.x), fmaf(u.y, cd.y, v.y * sd.y),              \
                           fmaf(u.z, cd.z, v.z * sd.z), fmaf(u.w, cd.w, v.w * sd.w));              \
        minus = make_float4(fmaf(u.x, cd.x, -(v.x * sd.x)), fmaf(u.y, cd.y, -(v.y * sd.y)),        \
                            fmaf(u.z, cd.z, -(v.z * sd.z)), fmaf(u.w, cd.w, -(v.w * sd.w)));       \
    }
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
        if (JET) {      /* table: cd_d = B_dj, sd0 = |B_j|^2 */                  \
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
        } else if (!(HALF)) {  /* sin rows: x + eps e_d -> s cd + c sd, x - eps e_d -> s cd - c sd */ \
            NSVD_STS(Bb_, rs);                                                   \
            if (DD > 0) { NSVD_PM(gp_, gm_, rs, rc, cd0, sd0) NSVD_STS(Bb_ + 32 * A_LD, gp_); NSVD_STS(Bb_ + 64 * A_LD, gm_); }   \
            if (DD > 1) { NSVD_PM(gp_, gm_, rs, rc, cd1, sd1) NSVD_STS(Bb_ + 96 * A_LD, gp_); NSVD_STS(Bb_ + 128 * A_LD, gm_); }  \
            if (DD > 2) { NSVD_PM(gp_, gm_, rs, rc, cd2, sd2) NSVD_STS(Bb_ + 160 * A_LD, gp_); NSVD_STS(Bb_ + 192 * A_LD, gm_); } \
        } else {        /* cos rows: x + eps e_d -> c cd - s sd, x - eps e_d -> c cd + s sd */ \
            NSVD_STS(Bb_, rc);                                                   \
            if (DD > 0) { NSVD_PM(gp_, gm_, rc, rs, cd0, sd0) NSVD_STS(Bb_ + 32 * A_LD, gm_); NSVD_STS(Bb_ + 64 * A_LD, gp_); }   \
            if (DD > 1) { NSVD_PM(gp_, gm_, rc, rs, cd1, sd1) NSVD_STS(Bb_ + 96 * A_LD, gm_); NSVD_STS(Bb_ + 128 * A_LD, gp_); }  \
            if (DD > 2) { NSVD_PM(gp_, gm_, rc, rs, cd2, sd2) NSVD_STS(Bb_ + 160 * A_LD, gm_); NSVD_STS(Bb_ + 192 * A_LD, gp_); } \
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
        if (DO_LOAD) NSVD_INTERLEAVE((PAR) ? 4 : 6 + 2 * DD, 0x020);                            \
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
#undef NSVD_LDG
#undef NSVD_STS

    }
    NSVD_STAMP(2)
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
        if (has_next) {
            const float* Wn = a.W[i + 1] + ((size_t)l * HID + 32 * w + hi) * HID;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = 2 * j + hi;
                nsvd_glds16(Wn + (size_t)(2 * j) * HID + 4 * (li ^ (row & 15)), Wt + 2 * j * HID);
            }
        }
        // softplus in registers; the centre rows' ACTIVATIONS are saved for the backward (its kernels then need
        // no softplus: sigmoid(z) = 1 - exp(-softplus(z)), and the weight gradients contract activations)
        float* zs = a.zsave[i] ? a.zsave[i] + ((size_t)l * HID + 32 * w) * a.B + b0 + li : nullptr;
        if (JET) {
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
        if (zs) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // DMA done, the 16 stores may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        NSVD_STAMP(5 + 4 * i)
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = (JET && e > 0) ? 0.f : nb[r];
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
    // ------------------------------------------------------------------ FD Hamiltonian epilogue
    // one thread per (stencil point, sample): output of the 128 -> 1 layer, then g_e = sqrt p(x_e) c base_e mask(x_e)
    // (the exp / sqrt heavy part, E x 32 threads wide instead of a 5-point loop on 32 threads)
    float* gs = outs;         // [NC]   g_e
    float* cen = red;         // [4][BS] centre: sqrt p, mask, |x|, base   (red is dead once read below)
    if (E == 1 && a.plain) {
        // model(x) = c * base * exp(-|x| / scales_l)   (reference pde/__init__.py:15-16), any input dimension
        if (tid < BS) {
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
    NsvdFdG og;
    float bve = 0.f;
    og.g = og.sp = og.mk = og.r = 0.f;
    const int e_t = tid / BS, sidx = tid - e_t * BS;
    if (tid < NC) {
        bve = (red[tid] + red[NC + tid]) + (red[2 * NC + tid] + red[3 * NC + tid]) + a.b[nh][l];
        float xc[NSVD_FD_MAXD];
        for (int d = 0; d < a.D; ++d) xc[d] = a.x[(size_t)(b0 + sidx) * a.D + d];
        const float s_l = a.scales ? a.scales[l] : 0.f;
        og = nsvd_fd_g(e_t, bve, xc, a.D, a.scales != nullptr, s_l, a.prob, a.log_norm);
    }
    __syncthreads();          // every thread has consumed its red[] inputs before the centre values overwrite them
    if (tid < NC) {
        gs[tid] = og.g;
        if (e_t == 0) {
            cen[sidx] = og.sp;
            cen[BS + sidx] = og.mk;
            cen[2 * BS + sidx] = og.r;
            cen[3 * BS + sidx] = bve;
        }
    }
    __syncthreads();
    NSVD_STAMP(13)
    if (tid < BS) {
        const int b = b0 + tid;
        float g[E];
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = gs[e * BS + tid];
        const float s_l = a.scales ? a.scales[l] : 0.f;
        const NsvdFdOut o = nsvd_fd_combine(g, cen[tid], cen[BS + tid], cen[2 * BS + tid], cen[3 * BS + tid], a.D,
                                            a.scales != nullptr, s_l, a.prob);
        const size_t idx = (size_t)b * a.L + l;
        a.f[idx] = o.f;
        a.Tf[idx] = o.Tf;
        if (a.jac) a.jac[idx] = o.jac;
        if (a.dsc) a.dsc[idx] = o.dsc;
    }
    NSVD_STAMP(14)
}

template <int E, int BF3 = 0>
size_t fwd_lds_bytes() {
    constexpr int NC = E * BS;
    constexpr int STAGE = 2 * HID * A_LD + 2 * NC * A_LD;
    constexpr int HSZ = NC * H_LD;
    constexpr int STAGE3 = (2 * 3 * (HID + NC) * B3_ROW) / 4;
    constexpr int RED_OFF = BF3 ? (STAGE3 > HSZ + HID * HID ? STAGE3 : HSZ + HID * HID)
                                : (STAGE > HSZ ? STAGE : HSZ) + HID * HID;
    return (RED_OFF + 5 * NC) * sizeof(float);
}

template <int E, int JET = 0, int BF3 = 0>
int launch_fwd(const FwdArgs& a, hipStream_t s) {
    const size_t lds = fwd_lds_bytes<E, BF3>();
    static bool attr_done = false;  // idempotent, racing threads set the same value
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)pmlp_fused_fwd_kernel<E, JET, BF3>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_done = true;
    }
    const int grid = (a.B / BS) * a.L;
    nsvd_prof_begin(s);
    hipLaunchKernelGGL((pmlp_fused_fwd_kernel<E, JET, BF3>), dim3(grid), dim3(256), lds, s, a);
    nsvd_prof_end(s);
    NSVD_CHECK_LAUNCH();
    return 0;
}

// ================================================================================================
// BACKWARD, part 1 (pmlp_fused_bwd_chain_kernel): one workgroup = head l x 32 centre samples.
// Data gradients only, walking from the output back to layer 0 with the forward's transposed tiles
// (rows = hidden units, columns = samples on the lanes; wave w owns rows 32w..32w+31):
//   dz_{nh-1} = W_last * dbase * sigmoid(z_{nh-1}),   dbase = df * d f / d base
//   dz_{i-1}[k][c] = (sum_n W_i[n][k] dz_i[n][c]) * sigmoid(z_{i-1}[k][c])     64 MFMAs/wave, W_i from L2
// Every dz_i (L, 128, B) goes to the workspace; all weight/bias gradients are reductions over the
// batch and are done by pmlp_fused_wgrad_kernel without atomics.
// This is autograd's backward of reference mlp.py:204-221 for the centre evaluation only (the 2D
// shifted evaluations carry no gradient: nestedlora.py:108-111) and of pde/__init__.py:16.
struct ChainArgs {
    const float* df;             // (B, L) d loss / d f, or null: derive it from the moments below
    const float* jac;
    const float* dsc;            // null without the exponential mask
    const float* W[NSVD_MAX_LAYERS];
    const float* zsave[NSVD_MAX_LAYERS];
    float* dz[NSVD_MAX_LAYERS];  // (L, 128, B) per hidden layer
    float* dbase;                // (L, B)  df * d f / d base    (for the last-layer gradient)
    float* dfsc;                 // (L, B)  df * d f / d scales  (exponential mask only)
    int nlayers, B, L;
    // EVD-loss mode (df == null): NestedLoRALossFunctionEVD.backward evaluated per sample right here
    NsvdEvdIn evd;
};

__global__ void __launch_bounds__(256, 1) pmlp_fused_bwd_chain_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float DZ[BS * H_LD];  // [c][n]  n contiguous
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nsb = a.B / BS;
    const int l = blockIdx.x / nsb;
    const int b0 = (blockIdx.x - l * nsb) * BS;
    const int nh = a.nlayers - 1;
    const int b = b0 + li;
    const size_t row0 = ((size_t)l * HID + 32 * w) * a.B + b;

    float dfv;
    if (a.df) {
        dfv = a.df[(size_t)b * a.L + l];
    } else {
        // d loss / d f[b][l] = gs * ( -(4/B) v_l Tf[b][l] + (2/B_half) sum_l' f[b][l'] M[l'][l] lam_other[l'][l] )
        // (reference methods/nestedlora.py:98-111 with f1, f2 = chunk(f, 2)); the moments are either the
        // reduced / all-reduced vector or this rank's per-chunk partial sums (reduced here, in a fixed order)
        float* col = DZ;  // [2][Lg] masked moment columns of (global) head lg (LDS scratch, free until the first exchange)
        const int Lg = a.evd.Lg, lg = a.evd.l_off + l;
        const int B1 = (a.B + 1) / 2, B2 = a.B - B1;
        if (a.evd.moments || a.evd.part) {
            for (int t = tid; t < 2 * Lg; t += 256) {
                const int h = t / Lg, lp = t - h * Lg;  // h = 0: lam_f1 column, 1: lam_f2 column
                col[t] = nsvd_evd_mask_M(a.evd, lp, lg, Lg) * nsvd_evd_lam(a.evd, h, lp * Lg + lg, a.B, Lg);
            }
            if (blockIdx.x == 0) nsvd_evd_finish(a.evd, a.B, Lg, DZ + 2 * Lg);
        } else {
            // direct mode: no moment kernel ran - the 2 Lg moments of THIS head's column straight from f
            // (8 threads per moment, fixed summation order; the loss scalars are not produced)
            const int sub = tid & 7;
            for (int t = tid >> 3; t < 2 * Lg; t += 32) {
                const int h = t / Lg, lp = t - h * Lg;
                const int r0 = h ? B1 : 0, nr = h ? B2 : B1;
                const float* fp = a.evd.f + (size_t)r0 * Lg;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int bb = sub;
                for (; bb + 24 < nr; bb += 32) {
                    s0 = fmaf(fp[(size_t)bb * Lg + lp], fp[(size_t)bb * Lg + lg], s0);
                    s1 = fmaf(fp[(size_t)(bb + 8) * Lg + lp], fp[(size_t)(bb + 8) * Lg + lg], s1);
                    s2 = fmaf(fp[(size_t)(bb + 16) * Lg + lp], fp[(size_t)(bb + 16) * Lg + lg], s2);
                    s3 = fmaf(fp[(size_t)(bb + 24) * Lg + lp], fp[(size_t)(bb + 24) * Lg + lg], s3);
                }
                for (; bb < nr; bb += 8) s0 = fmaf(fp[(size_t)bb * Lg + lp], fp[(size_t)bb * Lg + lg], s0);
                float sum = (s0 + s1) + (s2 + s3);
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 4, 64);
                if (sub == 0) col[t] = nsvd_evd_mask_M(a.evd, lp, lg, Lg) * (sum / (float)nr);
            }
        }
        __syncthreads();
        const bool first = b < B1;
        const float* cp = col + (first ? Lg : 0);  // the OTHER half's moments
        const float* fr = a.evd.f + (size_t)b * Lg;
        float acc = 0.f;
        for (int lp = 0; lp < Lg; ++lp) acc = fmaf(fr[lp], cp[lp], acc);
        dfv = a.evd.grad_scale * ((-4.f / (float)a.B) * nsvd_evd_mask_v(a.evd, lg, Lg) * a.evd.Tf[(size_t)b * Lg + lg] +
                                  (2.f / (float)(first ? B1 : B2)) * acc);
        __syncthreads();  // col[] is dead before DZ is reused
    }
    const float dbase = dfv * a.jac[(size_t)b * a.L + l];
    if (w == 0 && hi == 0) {
        a.dbase[(size_t)l * a.B + b] = dbase;
        if (a.dfsc) a.dfsc[(size_t)l * a.B + b] = dfv * a.dsc[(size_t)b * a.L + l];
    }
    float dz[16];
    {
        const float* zp = a.zsave[nh - 1] + row0;
        const float* wl = a.W[nh] + (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = acc_row(r, hi);
            dz[r] = wl[n] * dbase * nsvd_sigmoid_from_softplus(zp[(size_t)n * a.B]);
        }
    }
    for (int i = nh - 1; i >= 0; --i) {
        float* o = a.dz[i] + row0;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)acc_row(r, hi) * a.B] = dz[r];
        if (i == 0) break;
        // issue the loads the next tile needs before the LDS exchange: sigmoid inputs and W_i columns
        float zin[16];
        {
            const float* zp = a.zsave[i - 1] + row0;
#pragma unroll
            for (int r = 0; r < 16; ++r) zin[r] = zp[(size_t)acc_row(r, hi) * a.B];
        }
        __syncthreads();  // previous round's LDS reads are done
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(&DZ[li * H_LD + 32 * w + 8 * g + 4 * hi]) =
                make_float4(dz[4 * g], dz[4 * g + 1], dz[4 * g + 2], dz[4 * g + 3]);
        __syncthreads();
        f32x16 acc1[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
        const float* Wi = a.W[i] + (size_t)l * HID * HID + 32 * w + li;  // W_i[n][k = 32w + li]
        const float* Bp = DZ + li * H_LD + 4 * hi;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            Frag<1> f;
            const float* wp = Wi + (size_t)(8 * q + 4 * hi) * HID;
            f.a = make_float4(wp[0], wp[HID], wp[2 * HID], wp[3 * HID]);
            f.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * q);
            mma_frag<1>(acc1, f);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dz[r] = acc1[0][r] * nsvd_sigmoid_from_softplus(zin[r]);
    }
}

// ================================================================================================
// BACKWARD, part 2 (pmlp_fused_wgrad_kernel): every parameter gradient, one launch, no atomics.
// Three kinds of workgroup, told apart by blockIdx:
//   A  (F/128 * L):      dW_0[l][n][k] = sum_b dz_0[l][n][b] phi^T[k][b]: 128 x 128 tile, K = B, both
//                        operands b-contiguous, 4 waves as 2 x 2 of 64 x 64; the k = 0 tile of each head
//                        also writes db_0[l] (row sums of dz_0).
//   B  (4 (nh-1) L):     dW_i[l][n][k] = sum_b dz_i[l][n][b] softplus(z_{i-1}[l][k][b]), i >= 1: one
//                        64 x 64 quadrant of the 128 x 128 result, 4 waves of one 32 x 32 tile; quadrants
//                        in column 0 also write db_i. (128 x 64 half tiles were measured slower: fewer,
//                        longer workgroups next to the dW_0 tiles.)
//   C  (4 L):              dW_last[l][n] = sum_b dbase[b] softplus(z_{nh-1}[l][n][b]), db_last, d scales.
// Timeline at cfg2 (NSVD_WG_STAMPS build, scripts/dev_wgrad_stamps.py): A tiles run their K loop in 72-77 K
// cycles (65.5 K of MFMA issue) and end at 35 us; the B tiles, co-resident with them from t = 0 and latency-
// bound (single accumulator chain, softplus while staging), end at 49 us; the C tiles at 36 us. With the
// fused optimiser step the A epilogue moves 112 MB through HBM at once (20 us at 5.6 TB/s, all tiles finish
// together) and the B tail hides under it: 60 us, vs 50 + 20.5 us for separate backward and optimiser
// launches. Raising the B / C wave priority (s_setprio 3) shortens them but stretches the A loops by the same
// amount: no gain.
// K is streamed in 32-sample chunks through padded LDS tiles (rows of 36 floats, conflict-free
// ds_read_b128 fragments), register-staged and double buffered, one barrier per chunk.
struct WgradArgs {
    const float* dz[NSVD_MAX_LAYERS];     // (L, 128, B)
    const float* zsave[NSVD_MAX_LAYERS];  // (L, 128, B)
    const float* phiTc;                   // (F, B)
    const float* dbase;                   // (L, B) from the chain kernel
    const float* dfsc;                    // (L, B), null without the exponential mask
    float* gW[NSVD_MAX_LAYERS];           // gradients; may be null when the optimiser step is fused (opt != 0)
    float* gb[NSVD_MAX_LAYERS];
    float* gscales;
    int nlayers, B, L, F;
    int nA, nB;
    int bid0;  // first logical block of this launch (the A tiles and the B/C tiles are launched separately)
    // fused RMSprop + EMA epilogue (opt != 0): every gradient element is applied to its parameter in place
    // the moment it leaves the accumulator, so it never makes the HBM round trip (-8 B/parameter, -1 launch).
    // Safe in place: this kernel reads no parameter, the chain kernel that does has already run.
    int opt;
    NsvdHyper h;
    NsvdOptPtrs oW[NSVD_MAX_LAYERS], ob[NSVD_MAX_LAYERS], oscales;
    // split-K over the batch (S > 1; head-parallel ranks own few heads on many rows, so the tile count alone
    // would not fill the chip): tile (unit, slice) contracts rows [slice * Bs, (slice + 1) * Bs) and stores a
    // partial gradient into slice `slice` of `part`; wgrad_reduce_kernel adds the slices in order and stores the
    // gradient / takes the optimiser step.
    int S, Bs;
    float* part;
    size_t part_stride;                                        // floats per slice
    size_t poW[NSVD_MAX_LAYERS], pob[NSVD_MAX_LAYERS], poscales;  // tensor offsets inside a slice
};

// where a gradient element goes: the caller's gradient tensor (+ fused optimiser) or this slice's partial buffer
struct WgDst {
    float* g;
    int opt;
};
__device__ __forceinline__ WgDst wg_dst(const WgradArgs& a, float* g, size_t part_off, int slice) {
    if (a.S > 1) return WgDst{a.part + (size_t)slice * a.part_stride + part_off, 0};
    return WgDst{g, a.opt};
}

#ifdef NSVD_WG_STAMPS
// diagnostic build: per-block (kind, realtime start/end, cycles in prologue / loop / epilogue)
__device__ unsigned long long g_wg_stamps[1024 * 8];
#define WG_STAMP(slot, v) if (threadIdx.x == 0) g_wg_stamps[(size_t)blockIdx.x * 8 + (slot)] = (v)
#else
#define WG_STAMP(slot, v)
#endif

// one gradient element: store it and / or take the optimiser step on its parameter
__device__ __forceinline__ void wg_emit1(const WgradArgs& a, const WgDst& d, const NsvdOptPtrs& o, size_t off,
                                         float val) {
    float* g = d.g;
    if (g) g[off] = val;
    if (d.opt) {
        float pv = o.p[off], sv = o.sq[off], ev = o.ema ? o.ema[off] : 0.f;
        nsvd_rmsprop_upd(pv, val, sv, ev, o.ema != nullptr, a.h);
        o.p[off] = pv;
        o.sq[off] = sv;
        if (o.ema) o.ema[off] = ev;
    }
}

// the 16 accumulator registers of one 32 x 32 MFMA tile: rows acc_row(r, hi) * ld, this lane's column at `base`.
// Element offsets are 32-bit (every tensor is far below 2^32 bytes: checked on the host) so that the three state
// arrays share one offset register per element and the loads take the scalar-base form; with 64-bit offsets the 48
// loads in flight spill, and every spill waits for its load.
__device__ __forceinline__ float wg_ld(const float* p, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + byte_off);
}
__device__ __forceinline__ void wg_st(float* p, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(p) + byte_off) = v;
}

template <bool EMA>
__device__ __forceinline__ void wg_opt16(const WgradArgs& a, const NsvdOptPtrs& o, unsigned base, unsigned ld, int hi,
                                         const f32x16& acc) {
    float pv[16], sv[16], ev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // 48 independent loads in flight
        const unsigned off = 4u * (base + (unsigned)acc_row(r, hi) * ld);
        pv[r] = wg_ld(o.p, off);
        sv[r] = wg_ld(o.sq, off);
        ev[r] = EMA ? wg_ld(o.ema, off) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = 4u * (base + (unsigned)acc_row(r, hi) * ld);
        nsvd_rmsprop_upd(pv[r], acc[r], sv[r], ev[r], EMA, a.h);
        wg_st(o.p, off, pv[r]);
        wg_st(o.sq, off, sv[r]);
        if (EMA) wg_st(o.ema, off, ev[r]);
    }
}

__device__ __forceinline__ void wg_emit16(const WgradArgs& a, const WgDst& d, const NsvdOptPtrs& o, size_t base,
                                          size_t ld, int hi, const f32x16& acc) {
    float* g = d.g;
    if (g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wg_st(g, 4u * ((unsigned)base + (unsigned)acc_row(r, hi) * (unsigned)ld), acc[r]);
    }
    if (!d.opt) return;
    if (o.ema) wg_opt16<true>(a, o, (unsigned)base, (unsigned)ld, hi, acc);
    else wg_opt16<false>(a, o, (unsigned)base, (unsigned)ld, hi, acc);
}

// stage one 32-row x 32-column (float4 per thread) slab global -> registers
#define WG_LD(dst, src) dst = *reinterpret_cast<const float4*>(src)
#define WG_ST(dst, v) *reinterpret_cast<float4*>(dst) = (v)

__device__ __forceinline__ float4 softplus4(float4 v) {
    return make_float4(nsvd_softplus(v.x), nsvd_softplus(v.y), nsvd_softplus(v.z), nsvd_softplus(v.w));
}

__device__ __forceinline__ void wgrad_tile_A(const WgradArgs& a, float* As, float* Bs, int unit, int slice) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nkt = a.F / HID;
    const int l = unit / nkt;
    const int kf0 = (unit - l * nkt) * HID;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int s_row = tid >> 3, s_c4 = tid & 7;
    const float* a_src = a.dz[0] + ((size_t)l * HID + s_row) * a.B + (size_t)slice * a.Bs + 4 * s_c4;
    const float* b_src = a.phiTc + (size_t)(kf0 + s_row) * a.B + (size_t)slice * a.Bs + 4 * s_c4;
    const size_t step = (size_t)32 * a.B;
    // Same software pipeline as the forward's layer 0: fragments one q-group ahead, chunk c+1 written to
    // the other LDS buffer under chunk c's third q-group, chunk c+2 fetched from global under its fourth
    // (after the barrier), every memory instruction in an MFMA gap, branch-free steady state.
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;  // bias gradient partials (only used when kf0 == 0)
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define WA_LOAD(c)                                                 \
    {                                                              \
        const float* pa_ = a_src + (c) * BK;                       \
        const float* pb_ = b_src + (c) * BK;                       \
        WG_LD(ra0, pa_);                                           \
        WG_LD(ra1, pa_ + step);                                    \
        WG_LD(ra2, pa_ + 2 * step);                                \
        WG_LD(ra3, pa_ + 3 * step);                                \
        WG_LD(rb0, pb_);                                           \
        WG_LD(rb1, pb_ + step);                                    \
        WG_LD(rb2, pb_ + 2 * step);                                \
        WG_LD(rb3, pb_ + 3 * step);                                \
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
    const int nch = a.Bs / BK;
    WG_STAMP(0, 1ull);
    WG_STAMP(1, wall_clock64());
    WG_STAMP(2, __builtin_readcyclecounter());
    WA_LOAD(0);
    WA_STORE(0);
    __syncthreads();
    if (nch > 1) WA_LOAD(1);
    WG_STAMP(3, __builtin_readcyclecounter());
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
    WG_STAMP(4, __builtin_readcyclecounter());
    const size_t o = ((size_t)l * HID + 64 * wm) * a.F + kf0 + 64 * wn + li;
    const WgDst dW = wg_dst(a, a.gW[0], a.poW[0], slice), db = wg_dst(a, a.gb[0], a.pob[0], slice);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            wg_emit16(a, dW, a.oW[0], o + (size_t)(32 * i) * a.F + 32 * j, a.F, hi, acc[i][j]);
    if (kf0 == 0) {
        // bias gradient: 8 threads (s_c4) hold partial sums of rows s_row + {0, 32, 64, 96}
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            rs0 += __shfl_xor(rs0, off, 64);
            rs1 += __shfl_xor(rs1, off, 64);
            rs2 += __shfl_xor(rs2, off, 64);
            rs3 += __shfl_xor(rs3, off, 64);
        }
        if (s_c4 == 0) {
            const size_t gb = (size_t)l * HID + s_row;
            wg_emit1(a, db, a.ob[0], gb, rs0);
            wg_emit1(a, db, a.ob[0], gb + 32, rs1);
            wg_emit1(a, db, a.ob[0], gb + 64, rs2);
            wg_emit1(a, db, a.ob[0], gb + 96, rs3);
        }
    }
    WG_STAMP(5, __builtin_readcyclecounter());
    WG_STAMP(6, wall_clock64());
}

// dW_i quadrant through the shared C = A B^T tile routine (tile_nt.h): both operands are plain (L, 128, B) rows now
// that the forward saves activations - no softplus while staging, loads two chunks ahead, four accumulator chains.
// Needs the slice length to be a multiple of 64 (the 32-chunk form below takes the rest).
__device__ __forceinline__ void wgrad_tile_B64(const WgradArgs& a, float* lds, int unit, int slice) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int quad = unit & 3;
    const int rest = unit >> 2;
    const int l = rest % a.L;
    const int i = 1 + rest / a.L;
    const int n0 = (quad >> 1) * 64, k0 = (quad & 1) * 64;
    const float* A = a.dz[i] + ((size_t)l * HID + n0) * a.B;
    const float* Bm = a.zsave[i - 1] + ((size_t)l * HID + k0) * a.B;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    const int b0 = slice * a.Bs;
    if (k0 == 0) nsvd_tile_nt<true>(A, a.B, Bm, a.B, b0, b0 + a.Bs, lds, acc, rs);
    else nsvd_tile_nt<false>(A, a.B, Bm, a.B, b0, b0 + a.Bs, lds, acc, rs);
    wg_emit16(a, wg_dst(a, a.gW[i], a.poW[i], slice), a.oW[i],
              ((size_t)l * HID + n0 + 32 * (wv & 1)) * HID + k0 + 32 * (wv >> 1) + li, HID, hi, acc);
    if (k0 == 0) {
        // bias gradient: the 16 threads t & 15 of a staging row hold partial sums of rows (t >> 4) + 16 j
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rs[j] += __shfl_xor(rs[j], off, 64);
        if ((tid & 15) == 0) {
            const WgDst db = wg_dst(a, a.gb[i], a.pob[i], slice);
            const size_t gb = (size_t)l * HID + n0 + (tid >> 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) wg_emit1(a, db, a.ob[i], gb + 16 * j, rs[j]);
        }
    }
}

__device__ __forceinline__ void wgrad_tile_B(const WgradArgs& a, float* As, float* Bs, int unit, int slice) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nh = a.nlayers - 1;
    // unit -> (layer i in 1..nh-1, head l, quadrant)
    const int quad = unit & 3;
    const int rest = unit >> 2;
    const int l = rest % a.L;
    const int i = 1 + rest / a.L;
    (void)nh;
    const int n0 = (quad >> 1) * 64, k0 = (quad & 1) * 64;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int s_row = tid >> 3, s_c4 = tid & 7;  // 32 rows x 8 float4; two slabs per operand (64 rows)
    const float* a_src = a.dz[i] + ((size_t)l * HID + n0 + s_row) * a.B + (size_t)slice * a.Bs + 4 * s_c4;
    const float* b_src = a.zsave[i - 1] + ((size_t)l * HID + k0 + s_row) * a.B + (size_t)slice * a.Bs + 4 * s_c4;
    const size_t step = (size_t)32 * a.B;
    float4 pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
    float rs0 = 0.f, rs1 = 0.f;
    constexpr int TS = 64 * A_LD;  // one 64-row tile
#define WB_LOAD(S, c)                                 \
    {                                                 \
        WG_LD(S##a0, a_src + (c) * BK);               \
        WG_LD(S##a1, a_src + (c) * BK + step);        \
        WG_LD(S##b0, b_src + (c) * BK);               \
        WG_LD(S##b1, b_src + (c) * BK + step);        \
    }
#define WB_STORE(S, buf)                                                \
    {                                                                   \
        float* Ab_ = As + (buf) * TS + s_row * A_LD + 4 * s_c4;         \
        float* Bb_ = Bs + (buf) * TS + s_row * A_LD + 4 * s_c4;         \
        WG_ST(Ab_, S##a0);                                              \
        WG_ST(Ab_ + 32 * A_LD, S##a1);                                  \
        WG_ST(Bb_, S##b0);                                              \
        WG_ST(Bb_ + 32 * A_LD, S##b1);                                  \
        rs0 += (S##a0.x + S##a0.y) + (S##a0.z + S##a0.w);               \
        rs1 += (S##a1.x + S##a1.y) + (S##a1.z + S##a1.w);               \
    }
#define WB_COMPUTE(cur)                                                                        \
    {                                                                                          \
        const float* Ap = As + (cur) * TS + (32 * wm + li) * A_LD + 4 * hi;                    \
        const float* Bp = Bs + (cur) * TS + (32 * wn + li) * A_LD + 4 * hi;                    \
        _Pragma("unroll") for (int q = 0; q < BK / 8; ++q) {                                   \
            const float4 av = *reinterpret_cast<const float4*>(Ap + 8 * q);                    \
            const float4 bv = *reinterpret_cast<const float4*>(Bp + 8 * q);                    \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);              \
        }                                                                                      \
    }
    const int nch = a.Bs / BK;
    pa0 = pa1 = pb0 = pb1 = qa0 = qa1 = qb0 = qb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    WB_LOAD(p, 0);
    if (nch > 1) WB_LOAD(q, 1);
    WB_STORE(p, 0);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
        if (c + 2 < nch) WB_LOAD(p, c + 2);
        WB_COMPUTE(0);
        if (c + 1 < nch) WB_STORE(q, 1);
        __syncthreads();
        if (c + 1 < nch) {
            if (c + 3 < nch) WB_LOAD(q, c + 3);
            WB_COMPUTE(1);
            if (c + 2 < nch) WB_STORE(p, 0);
            __syncthreads();
        }
    }
#undef WB_COMPUTE
#undef WB_LOAD
#undef WB_STORE
    wg_emit16(a, wg_dst(a, a.gW[i], a.poW[i], slice), a.oW[i],
              ((size_t)l * HID + n0 + 32 * wm) * HID + k0 + 32 * wn + li, HID, hi, acc);
    if (k0 == 0) {
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            rs0 += __shfl_xor(rs0, off, 64);
            rs1 += __shfl_xor(rs1, off, 64);
        }
        if (s_c4 == 0) {
            const size_t gb = (size_t)l * HID + n0 + s_row;
            const WgDst db = wg_dst(a, a.gb[i], a.pob[i], slice);
            wg_emit1(a, db, a.ob[i], gb, rs0);
            wg_emit1(a, db, a.ob[i], gb + 32, rs1);
        }
    }
}

__device__ __forceinline__ void wgrad_tile_C(const WgradArgs& a, float* lds, int unit, int slice) {
    const int l = unit >> 2, part = unit & 3;  // 4 workgroups per head: 32 of the 128 rows each
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int nh = a.nlayers - 1;
    float* red = lds;            // [8]
    float* dbl = lds + 16;       // [Bs] dbase of this head (Bs <= 2 * 128 * 36 - 16 floats, checked on the host)
    const size_t row0 = (size_t)l * a.B + (size_t)slice * a.Bs;  // this slice's rows of the head's (B) vectors
    float sb = 0.f, ss = 0.f;
    for (int b = tid; b < a.Bs; b += 256) {
        const float v = a.dbase[row0 + b];
        dbl[b] = v;
        sb += v;
        if (a.dfsc) ss += a.dfsc[row0 + b];
    }
    sb = nsvd_wave_sum(sb);
    ss = nsvd_wave_sum(ss);
    if (lane == 0) {
        red[w] = sb;
        red[4 + w] = ss;
    }
    __syncthreads();
    if (tid == 0 && part == 0) {
        wg_emit1(a, wg_dst(a, a.gb[nh], a.pob[nh], slice), a.ob[nh], l, (red[0] + red[1]) + (red[2] + red[3]));
        if (a.dfsc)
            wg_emit1(a, wg_dst(a, a.gscales, a.poscales, slice), a.oscales, l,
                     (red[4] + red[5]) + (red[6] + red[7]));
    }
    // dW_last[n] = sum_b dbase[b] softplus(z[n][b]): each wave owns 32 rows and walks them 8 at a time so
    // that 8 independent 16-B loads are in flight per lane (a row-at-a-time loop is pure L2 latency)
    {
        const int n0 = 32 * part + 8 * w;  // this wave's 8 rows
        float s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
        const float* zrow = a.zsave[nh - 1] + (size_t)slice * a.Bs;
        for (int b = 4 * lane; b < a.Bs; b += 256) {
            float4 z[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                z[j] = *reinterpret_cast<const float4*>(zrow + ((size_t)l * HID + n0 + j) * a.B + b);
            const float4 d = *reinterpret_cast<const float4*>(dbl + b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] = fmaf(d.x, z[j].x, s[j]);
                s[j] = fmaf(d.y, z[j].y, s[j]);
                s[j] = fmaf(d.z, z[j].z, s[j]);
                s[j] = fmaf(d.w, z[j].w, s[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = nsvd_wave_sum(s[j]);
            if (lane == 0)
                wg_emit1(a, wg_dst(a, a.gW[nh], a.poW[nh], slice), a.oW[nh], (size_t)l * HID + n0 + j, t);
        }
    }
}

__global__ void __launch_bounds__(256, 2) pmlp_fused_wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float smem_wg[4 * HID * A_LD];  // 72 KB: two blocks per CU
    static_assert(4 * HID * A_LD >= NSVD_TNT_FLOATS, "tile_nt buffers must fit the weight-gradient LDS");
    float* As = smem_wg;
    float* Bs = smem_wg + 2 * HID * A_LD;
    // grid = S x (nA | nB | 4 L) blocks, kind-major so that the long dW_0 tiles are dispatched first
    int bid = blockIdx.x + a.bid0;
    if (bid < a.nA * a.S) {
        const int slice = bid / a.nA;
        bid -= slice * a.nA;
        // heads share an XCD (dz_0[l] stays in that L2) when the tile count allows the remap
        int unit = bid;
        if ((a.nA & 7) == 0) unit = (bid & 7) * (a.nA >> 3) + (bid >> 3);
        wgrad_tile_A(a, As, Bs, unit, slice);
        return;
    }
    bid -= a.nA * a.S;
    if (bid < a.nB * a.S) {
        WG_STAMP(0, 2ull);
        WG_STAMP(1, wall_clock64());
        if (a.Bs % NSVD_TNT_KC == 0) wgrad_tile_B64(a, smem_wg, bid % a.nB, bid / a.nB);
        else wgrad_tile_B(a, As, Bs, bid % a.nB, bid / a.nB);
        WG_STAMP(6, wall_clock64());
        return;
    }
    bid -= a.nB * a.S;
    WG_STAMP(0, 3ull);
    WG_STAMP(1, wall_clock64());
    wgrad_tile_C(a, As, bid % (4 * a.L), bid / (4 * a.L));
    WG_STAMP(6, wall_clock64());
}

// Split-K second pass: gradient = sum of the S partial slices (in slice order), then stored and / or applied
// (RMSprop + EMA) exactly as the S = 1 epilogue does. One launch over all tensors of the model.
struct ReduceArgs {
    const float* part;
    size_t part_stride;
    int S, ntensors, opt;
    NsvdHyper h;
    size_t off[2 * NSVD_MAX_LAYERS + 1], n[2 * NSVD_MAX_LAYERS + 1];  // slice offset / element count (multiples of 4)
    float* g[2 * NSVD_MAX_LAYERS + 1];
    NsvdOptPtrs o[2 * NSVD_MAX_LAYERS + 1];
    size_t total4;  // float4 groups over all tensors
};

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(ReduceArgs a) {
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < a.total4; q += (size_t)gridDim.x * 256) {
        // tensors are laid out back to back (padded to 4 floats) inside a slice: find the one holding group q
        int t = 0;
        size_t e = q * 4;
        while (t + 1 < a.ntensors && e >= a.off[t + 1]) ++t;
        e -= a.off[t];
        if (e >= a.n[t]) continue;  // alignment padding between tensors
        const float* src = a.part + a.off[t] + e;
        float4 gsum = *reinterpret_cast<const float4*>(src);
        for (int sl = 1; sl < a.S; ++sl) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sl * a.part_stride);
            gsum.x += v.x; gsum.y += v.y; gsum.z += v.z; gsum.w += v.w;
        }
        const int cnt = a.n[t] - e < 4 ? (int)(a.n[t] - e) : 4;
        const float gv[4] = {gsum.x, gsum.y, gsum.z, gsum.w};
        for (int c = 0; c < cnt; ++c) {
            if (a.g[t]) a.g[t][e + c] = gv[c];
            if (a.opt) {
                const NsvdOptPtrs& o = a.o[t];
                float pv = o.p[e + c], sv = o.sq[e + c], ev = o.ema ? o.ema[e + c] : 0.f;
                nsvd_rmsprop_upd(pv, gv[c], sv, ev, o.ema != nullptr, a.h);
                o.p[e + c] = pv;
                o.sq[e + c] = sv;
                if (o.ema) o.ema[e + c] = ev;
            }
        }
    }
}
#undef WG_LD
#undef WG_ST

struct FusedWs {
    float* phi;                       // (B, F) sample-major Fourier features of the centre rows
    float* sctab;                     // (D, 2, m) cos / sin of eps * fourier_B (stencil rows by angle addition)
    float* phiTc;                     // (F, B) feature-major copy of the centre rows (weight gradient)
    float* zsave[NSVD_MAX_LAYERS];    // (L, 128, B) per hidden layer
    float* jac;                       // (B, L)
    float* dsc;                       // (B, L)
    float* dz[NSVD_MAX_LAYERS];       // (L, 128, B) per hidden layer
    float* dbase;                     // (L, B)
    float* dfsc;                      // (L, B)
    float* gpart;                     // (S, slice) split-K partial gradients, S = wgrad_slices() > 1 only
    unsigned short* w0p;              // (3, L, 128, F) bf16 planes of W_0 (NSVD_PATH_FUSED_BF16X3 only)
    size_t bytes;
};

// Batch slices of the weight-gradient contraction: doubled while the dW_0 tile count leaves CUs idle and a slice
// keeps at least 256 rows (8 chunks) to amortise a tile's prologue and epilogue.
int wgrad_slices(const nsvd_model_desc& d, int B) {
    const int nA = (2 * d.m / HID) * d.L;
    int S = 1;
    while (S < 16 && nA * S < 256 && (B / (2 * S)) % BK == 0 && B / (2 * S) >= 256) S *= 2;
    return S;
}

// per-slice layout of the partial gradients: [W_0 | .. | W_n | b_0 | .. | b_n | scales], each padded to 4 floats
struct PartLayout {
    size_t oW[NSVD_MAX_LAYERS], ob[NSVD_MAX_LAYERS], oscales, nW[NSVD_MAX_LAYERS], nb[NSVD_MAX_LAYERS], nscales;
    size_t stride;
};
PartLayout part_layout(const nsvd_model_desc& d) {
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

FusedWs carve_fused(const nsvd_model_desc& d, int B, void* base) {
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
    w.sctab = take((size_t)2 * d.D * d.m);
    w.phiTc = take(F * B);
    for (int i = 0; i < d.nlayers - 1; ++i) w.zsave[i] = take((size_t)d.L * HID * B);
    w.jac = take((size_t)B * d.L);
    w.dsc = take((size_t)B * d.L);
    for (int i = 0; i < d.nlayers - 1; ++i) w.dz[i] = take((size_t)d.L * HID * B);
    w.dbase = take((size_t)B * d.L);
    w.dfsc = take((size_t)B * d.L);
    const int S = wgrad_slices(d, B);
    w.gpart = S > 1 ? take((size_t)S * part_layout(d).stride) : nullptr;
    w.w0p = (unsigned short*)take(((size_t)3 * d.L * HID * F + 1) / 2);
    w.bytes = off;
    return w;
}

}  // namespace

bool nsvd_fused_supported(const nsvd_model_desc& d, int B, bool exact) {
    // E <= 5 columns per sample: the forward's LDS image (155 KB at E = 5). Stencil: E = 1 + 2D, so D <= 2; the
    // exact-Laplacian jets have E = D + 2, so D <= 3.
    if (d.D < 1 || d.D > (exact ? 3 : 2)) return false;
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
    {
        const int nsb = B / BS;
        double best = 1e300;
        a.xcd_remap = 0;
        for (int HX = 1; HX <= 8; HX *= 2) {
            const int SX = 8 / HX;
            if (d.L % HX != 0 || nsb % SX != 0) continue;
            // (the feature slab is the centre rows only: the stencil rows are generated in the kernel)
            const double bytes = (double)(d.L / HX) * HID * F + (double)(nsb / SX) * BS * F;
            if (bytes < best) {
                best = bytes;
                a.xcd_remap = HX;
            }
        }
    }
#ifdef NSVD_FWD_STAMPS
    a.stamps = (unsigned long long*)w.dz[0];  // diagnostic build: stamps land in the (then unused) dz_0 scratch
#endif
    if (bf3) {  // opt-in: layer 0 on the bf16 MFMA with three-way split operands
        const size_t n4 = (size_t)d.L * HID * F / 4;
        hipLaunchKernelGGL(w0_split_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const float4*>(p.W[0]),
                           reinterpret_cast<uint2*>(w.w0p), n4);
        NSVD_CHECK_LAUNCH();
        a.w0p = w.w0p;
        a.w0_plane = (size_t)d.L * HID * F;
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
    switch (E) {
        case 3: return launch_fwd<3>(a, s);
        case 5: return launch_fwd<5>(a, s);
    }
    return NSVD_EUNSUPPORTED;
}

// ---- plain model evaluation on the fused kernels (E = 1): out = c * model(x), any input dimension --------------
bool nsvd_fused_model_supported(const nsvd_model_desc& d, int B) {
    if (d.D < 1 || d.D > 64) return false;
    if (d.nlayers < 2) return false;
    for (int i = 0; i < d.nlayers - 1; ++i)
        if (d.dims[i] != HID) return false;
    if (B % BS != 0 || B > 65536) return false;
    if (B / wgrad_slices(d, B) > 8192) return false;
    if ((2 * d.m) % HID != 0) return false;
    return true;
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
    {
        const int nsb = B / BS;
        double best = 1e300;
        for (int HX = 1; HX <= 8; HX *= 2) {
            const int SX = 8 / HX;
            if (d.L % HX != 0 || nsb % SX != 0) continue;
            const double bytes = (double)(d.L / HX) * HID * F + (double)(nsb / SX) * BS * F;
            if (bytes < best) {
                best = bytes;
                a.xcd_remap = HX;
            }
        }
    }
    return launch_fwd<1>(a, s);
}

static int fused_backward_impl(const nsvd_model_desc& d, const nsvd_params& p, int B, const float* df,
                               const NsvdEvdIn* evd, const nsvd_params* gp, const NsvdOptStep* opt, void* ws,
                               hipStream_t s) {
    nsvd_params g;
    memset(&g, 0, sizeof(g));
    if (gp) g = *gp;
    const FusedWs w = carve_fused(d, B, ws);
    const int F = 2 * d.m, nh = d.nlayers - 1;

    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.df = df;
    if (evd) a.evd = *evd;
    a.jac = w.jac;
    a.dsc = d.has_exp_mask ? w.dsc : nullptr;
    a.dbase = w.dbase;
    a.dfsc = d.has_exp_mask ? w.dfsc : nullptr;
    for (int i = 0; i < d.nlayers; ++i) {
        a.W[i] = p.W[i];
        a.zsave[i] = (i < nh) ? w.zsave[i] : nullptr;
        a.dz[i] = (i < nh) ? w.dz[i] : nullptr;
    }
    a.nlayers = d.nlayers; a.B = B; a.L = d.L;
    hipLaunchKernelGGL(pmlp_fused_bwd_chain_kernel, dim3((B / BS) * d.L), dim3(256), 0, s, a);
    NSVD_CHECK_LAUNCH();

    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    for (int i = 0; i < d.nlayers; ++i) {
        wa.dz[i] = (i < nh) ? w.dz[i] : nullptr;
        wa.zsave[i] = (i < nh) ? w.zsave[i] : nullptr;
        wa.gW[i] = g.W[i];
        wa.gb[i] = g.b[i];
    }
    wa.phiTc = w.phiTc;
    wa.dbase = w.dbase;
    wa.dfsc = d.has_exp_mask ? w.dfsc : nullptr;
    wa.gscales = d.has_exp_mask ? g.scales : nullptr;
    wa.nlayers = d.nlayers; wa.B = B; wa.L = d.L; wa.F = F;
    if (opt) {
        wa.opt = 1;
        wa.h = opt->h;
        for (int i = 0; i < d.nlayers; ++i) {
            wa.oW[i] = NsvdOptPtrs{p.W[i], opt->sq.W[i], opt->ema ? opt->ema->W[i] : nullptr};
            wa.ob[i] = NsvdOptPtrs{p.b[i], opt->sq.b[i], opt->ema ? opt->ema->b[i] : nullptr};
        }
        if (d.has_exp_mask) wa.oscales = NsvdOptPtrs{p.scales, opt->sq.scales, opt->ema ? opt->ema->scales : nullptr};
    }
    wa.nA = (F / HID) * d.L;
    wa.nB = 4 * (nh - 1) * d.L;
    wa.S = wgrad_slices(d, B);
    wa.Bs = B / wa.S;
    const PartLayout pl = part_layout(d);
    if (wa.S > 1) {
        wa.part = w.gpart;
        wa.part_stride = pl.stride;
        for (int i = 0; i < d.nlayers; ++i) {
            wa.poW[i] = pl.oW[i];
            wa.pob[i] = pl.ob[i];
        }
        wa.poscales = pl.oscales;
    }
    // One launch: the dW_0 tiles go one per CU first, the small dW_i / db / last-layer workgroups
    // then co-reside with them (measured: 55 us together vs 42 + 23 us as two launches).
    wa.bid0 = 0;
    hipLaunchKernelGGL(pmlp_fused_wgrad_kernel, dim3(wa.S * (wa.nA + wa.nB + 4 * d.L)), dim3(256), 0, s, wa);
    NSVD_CHECK_LAUNCH();
    if (wa.S == 1) return 0;
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.part = w.gpart;
    ra.part_stride = pl.stride;
    ra.S = wa.S;
    ra.opt = wa.opt;
    ra.h = wa.h;
    int t = 0;
    for (int i = 0; i < d.nlayers; ++i, ++t) {
        ra.off[t] = pl.oW[i]; ra.n[t] = pl.nW[i]; ra.g[t] = g.W[i]; ra.o[t] = wa.oW[i];
    }
    for (int i = 0; i < d.nlayers; ++i, ++t) {
        ra.off[t] = pl.ob[i]; ra.n[t] = pl.nb[i]; ra.g[t] = g.b[i]; ra.o[t] = wa.ob[i];
    }
    if (d.has_exp_mask) {
        ra.off[t] = pl.oscales; ra.n[t] = pl.nscales; ra.g[t] = g.scales; ra.o[t] = wa.oscales;
        ++t;
    }
    ra.ntensors = t;
    ra.total4 = (pl.oscales + (pl.nscales + 3) / 4 * 4) / 4;
    size_t blocks = (ra.total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ra);
    NSVD_CHECK_LAUNCH();
    return 0;
}

#ifdef NSVD_WG_STAMPS
extern "C" int nsvd_debug_wgrad_stamps(unsigned long long* host, size_t n) {
    return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), n * sizeof(unsigned long long));
}
#endif

int nsvd_fused_backward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                        int B, const float* df, const nsvd_params& g, void* ws, hipStream_t s) {
    (void)prob;
    (void)x;
    return fused_backward_impl(d, p, B, df, nullptr, &g, nullptr, ws, s);
}

int nsvd_fused_backward_evd(const nsvd_model_desc& d, const nsvd_params& p, int B, const NsvdEvdIn& evd,
                            const nsvd_params* g, const NsvdOptStep* opt, void* ws, hipStream_t s) {
    if (!g && !opt) return NSVD_EINVAL;
    return fused_backward_impl(d, p, B, nullptr, &evd, g, opt, ws, s);
}
