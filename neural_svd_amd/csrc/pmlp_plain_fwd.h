// Plain model evaluation in STREAMING form (pmlp_plain_stream_fwd_kernel) for the kernel-operator row's model shape
// (F = 2 m = 128 features, two hidden layers of 128): included by pmlp_fwd.hip.
//
// The tile kernel (pmlp_fused_fwd_kernel<E, 0, 0, 1>) re-stages a head's W_0 / W_1 for every 128 samples and runs one
// wave per SIMD, whose softplus phases do not hide under its own MFMAs (0.54 of the fp32 MFMA peak at configs[3]). Here
// a workgroup owns one head and a run of 32-sample tiles with BOTH weight matrices resident in registers as MFMA A
// fragments (2 x 64 registers per lane), the features of a tile are its only staged operand (16 KB, double buffered),
// and at 51 KB of LDS and < 256 registers two workgroups share a CU: one's softplus / stores run under the other's MFMAs.
//   per tile:  z_0 = b_0 + W_0 phi^T   64 MFMAs / wave   -> a_0 = softplus -> saved, LDS exchange
//              z_1 = b_1 + W_1 a_0     64 MFMAs / wave   -> a_1 = softplus -> saved
//              out = c (b_2 + W_2 a_1) mask               register dot + cross-wave LDS sum
// The arithmetic (MFMA order over k, softplus, epilogue) is the tile kernel's: same bits.
// Reference: examples/models/mlp.py:204-221, examples/operator/pde/__init__.py:15-16 (as pmlp_fwd.hip).
#pragma once

struct PlainFwdArgs {
    const float* phi;     // (B, 128) features, sample-major
    const float *W0, *b0, *W1, *b1, *Wl, *bl;  // (L,128,128) (L,128) (L,128,128) (L,128) (L,128) (L)
    const float* x;       // (B, D), read only with the exponential mask
    const float* scales;  // (L) or null
    int D;
    float c;
    float* out;           // (B, L)
    float* jac;           // (B, L) or null
    float* dsc;           // (B, L) or null
    float* z0;            // (L, 128, B) saved activations, or null
    float* z1;
    int B, L, tpw;        // tiles (of 32 samples) per workgroup
};

constexpr int PF_LD = HID + 4;                                  // padded row (floats) of the [sample][k] images
constexpr int PF_LDS_FLOATS = 3 * BS * PF_LD + 4 * BS;          // phi x 2, activations, reduction scratch
constexpr size_t PF_LDS_BYTES = (size_t)PF_LDS_FLOATS * sizeof(float);

__global__ void __launch_bounds__(256, 2) pmlp_plain_stream_fwd_kernel(PlainFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pf_smem[];
    float* PH = pf_smem;                    // [2][32 samples][132]
    float* Hs = PH + 2 * BS * PF_LD;        // [32 samples][132]  a_0, k contiguous
    float* red = Hs + BS * PF_LD;           // [4 waves][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nrun = (a.B / BS) / a.tpw;    // runs of tiles per head
    const int l = blockIdx.x / nrun;
    const int t0 = (blockIdx.x - l * nrun) * a.tpw;

    // ---- resident operands: A fragments of both layers (W[n = 32 w + li][k = 8 q + 4 hi + j] at 4 q + j), biases, the
    // 128 -> 1 weights of this wave's rows
    float WA[64], WB[64], b0v[16], b1v[16], wlv[16];
    {
        const float* p0 = a.W0 + ((size_t)l * HID + 32 * w + li) * HID + 4 * hi;
        const float* p1 = a.W1 + ((size_t)l * HID + 32 * w + li) * HID + 4 * hi;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 u = *reinterpret_cast<const float4*>(p0 + 8 * q);
            const float4 v = *reinterpret_cast<const float4*>(p1 + 8 * q);
            WA[4 * q] = u.x; WA[4 * q + 1] = u.y; WA[4 * q + 2] = u.z; WA[4 * q + 3] = u.w;
            WB[4 * q] = v.x; WB[4 * q + 1] = v.y; WB[4 * q + 2] = v.z; WB[4 * q + 3] = v.w;
        }
        const size_t rb = (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            b0v[r] = a.b0[rb + acc_row(r, hi)];
            b1v[r] = a.b1[rb + acc_row(r, hi)];
            wlv[r] = a.Wl[rb + acc_row(r, hi)];
        }
    }
    const float blv = a.bl[l];

    // a tile of features is 32 rows x 512 B, contiguous in memory: thread t moves float4 t + 256 k
    float4 pre[4];
    auto request = [&](int t) {
        const float4* src = reinterpret_cast<const float4*>(a.phi + (size_t)(t0 + t) * BS * HID) + tid;
#pragma unroll
        for (int k = 0; k < 4; ++k) pre[k] = src[256 * k];
    };
    request(0);
#define PF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    for (int t = 0; t < a.tpw; ++t) {
        float* P = PH + (t & 1) * BS * PF_LD;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k, row = idx >> 5, c4 = idx & 31;
            *reinterpret_cast<float4*>(P + row * PF_LD + 4 * c4) = pre[k];
        }
        if (t + 1 < a.tpw) request(t + 1);
        PF_BARRIER();  // (1) the tile's features in place; the previous tile's readers of Hs / red are done
        const int b0 = (t0 + t) * BS;
        f32x16 acc[1];
        // ---------------------------------------------------------------- layer 0
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = b0v[r];
        {
            const float* Bp = P + li * PF_LD + 4 * hi;
            Frag<1> g0, g1;
            g0.b[0] = *reinterpret_cast<const float4*>(Bp);
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                g1.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * (q + 1));
                g0.a = make_float4(WA[4 * q], WA[4 * q + 1], WA[4 * q + 2], WA[4 * q + 3]);
                mma_frag<1>(acc, g0);
                if (q + 2 < 16) g0.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * (q + 2));
                g1.a = make_float4(WA[4 * q + 4], WA[4 * q + 5], WA[4 * q + 6], WA[4 * q + 7]);
                mma_frag<1>(acc, g1);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = nsvd_softplus(acc[0][r]);
        if (a.z0) {
            float* zs = a.z0 + ((size_t)l * HID + 32 * w) * a.B + b0 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) zs[(size_t)acc_row(r, hi) * a.B] = acc[0][r];
        }
        {   // registers 4 g .. 4 g + 3 of a lane are 4 consecutive hidden rows 8 g + 4 hi + (0..3): one 16-B store
            float* hcol = Hs + li * PF_LD + 32 * w + 4 * hi;
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(hcol + 8 * g) =
                    make_float4(acc[0][4 * g], acc[0][4 * g + 1], acc[0][4 * g + 2], acc[0][4 * g + 3]);
        }
        PF_BARRIER();  // (2) a_0 complete
        // ---------------------------------------------------------------- layer 1
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = b1v[r];
        {
            const float* Bp = Hs + li * PF_LD + 4 * hi;
            Frag<1> g0, g1;
            g0.b[0] = *reinterpret_cast<const float4*>(Bp);
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                g1.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * (q + 1));
                g0.a = make_float4(WB[4 * q], WB[4 * q + 1], WB[4 * q + 2], WB[4 * q + 3]);
                mma_frag<1>(acc, g0);
                if (q + 2 < 16) g0.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * (q + 2));
                g1.a = make_float4(WB[4 * q + 4], WB[4 * q + 5], WB[4 * q + 6], WB[4 * q + 7]);
                mma_frag<1>(acc, g1);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = nsvd_softplus(acc[0][r]);
        if (a.z1) {
            float* zs = a.z1 + ((size_t)l * HID + 32 * w) * a.B + b0 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) zs[(size_t)acc_row(r, hi) * a.B] = acc[0][r];
        }
        // ---------------------------------------------------------------- 128 -> 1 layer
        {
            float part = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) part = fmaf(wlv[r], acc[0][r], part);
            part += __shfl_xor(part, 32, 64);
            if (hi == 0) red[w * BS + li] = part;
        }
        PF_BARRIER();  // (3)
        if (tid < BS) {
            // model(x) = c * base * exp(-|x| / scales_l)   (reference pde/__init__.py:15-16), any input dimension
            const int b = b0 + tid;
            const float bv = (red[tid] + red[BS + tid]) + (red[2 * BS + tid] + red[3 * BS + tid]) + blv;
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
            a.out[idx] = a.c * bv * mk;
            if (a.jac) a.jac[idx] = a.c * mk;
            if (a.dsc) a.dsc[idx] = a.scales ? a.c * bv * mk * r / (s_l * s_l) : 0.f;
        }
    }
#undef PF_BARRIER
}

// tiles per workgroup: runs as long as possible while the grid keeps two workgroups on every CU (0: not this kernel)
inline int plain_stream_tpw(const nsvd_model_desc& d, int B) {
    if (d.nlayers != 3 || 2 * d.m != HID || d.dims[0] != HID || d.dims[1] != HID || B % BS != 0) return 0;
    const int nt = B / BS;
    if ((long)nt * d.L < 1024) return 0;
    int tpw = 1;
    while (tpw < 64 && nt % (2 * tpw) == 0 && (long)(nt / (2 * tpw)) * d.L >= 512) tpw *= 2;
    return tpw;
}
