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
// Plane layout: CHUNK-MAJOR, (3, L, F/32 chunks, 128 rows, 32 k), chunks in the order the K loop visits them (pair j
// of 32-wide sin / cos chunks: k = 32 j .. and m + 32 j ..). The 8 KB tile of a chunk is then contiguous - a wave's
// load instruction covers 8 full 128-byte lines instead of the halves of 16 lines whose other halves belong to a
// chunk two steps away.
__global__ void __launch_bounds__(256) w0_split_kernel(const float4* __restrict__ W, uint2* __restrict__ P, int L, int m) {
    const int F = 2 * m, q4 = F / 4;
    const size_t n4 = (size_t)L * HID * q4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const int k = 4 * (int)(i % q4);
        const size_t row = i / q4;  // l * 128 + n
        const int n = (int)(row % HID), l = (int)(row / HID);
        const int half = k >= m, kk = k - half * m, c = 2 * (kk / BK) + half, within = kk % BK;
        const size_t o = ((((size_t)l * (F / BK) + c) * HID + n) * BK + within) / 4;  // in units of 4 elements
        uint2 p0, p1, p2;
        nsvd_bf3_split(W[i], p0, p1, p2);
        P[o] = p0;
        P[n4 + o] = p1;
        P[2 * n4 + o] = p2;
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
    // W tile of a chunk, per plane 128 rows x 64 B, contiguous in the chunk-major plane layout: piece q (16 B) of row r
    // for thread index r * 4 + q (2 per thread: rows r and r + 64)
    const int w_row = tid >> 2, w_q = tid & 3;
    const unsigned short* w_src = a.w0p + (size_t)l * HID * a.F + 8 * tid;
    const size_t w_half = (size_t)64 * BK;  // second piece: row + 64

    auto load = [&](int c, auto half) {  // chunk c = pair (c >> 1), half (c & 1)
        constexpr int HALF = decltype(half)::value;
        const int kp = (c >> 1) * BK;
        const unsigned short* pw = w_src + (size_t)c * (HID * BK);
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
