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
#include <string.h>
#include "nsvd_kernels.h"
#include "fd_math.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HID = 128;      // hidden width
constexpr int BS = 32;        // base samples per workgroup
constexpr int BK = 32;        // layer-0 K chunk
constexpr int A_LD = BK + 4;  // padded row of the W_0 tile (floats)

struct FwdArgs {
    const float* phiT;
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
    float* zsave[NSVD_MAX_LAYERS];  // (L, 128, B) per hidden layer, or null
    int xcd_remap;
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

template <int E>
__global__ void __launch_bounds__(256, 1) pmlp_fused_fwd_kernel(FwdArgs a) {
    constexpr int NC = E * BS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][128][A_LD]   W_0 tile, k contiguous
    float* Bs = smem + 2 * HID * A_LD;      // [2][NC][A_LD]    phi tile (rows = sample columns), k contiguous
    float* Hs = smem;                       // [NC][H_LD]       activations, k contiguous (aliases the stage buffers)
    constexpr int STAGE = 2 * HID * A_LD + 2 * NC * A_LD;
    constexpr int HSZ = NC * H_LD;
    float* red = smem + (STAGE > HSZ ? STAGE : HSZ);  // [4][NC]
    float* outs = red + 4 * NC;                       // [NC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nsb = a.B / BS;
    int unit = blockIdx.x;
    if (a.xcd_remap) {
        const int per = gridDim.x >> 3;
        unit = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
    const int l = unit / nsb;
    const int b0 = (unit - l * nsb) * BS;

    f32x16 acc[E];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;

    // ------------------------------------------------------------------ layer 0: K = F in chunks of BK
    // Both operands are k-contiguous rows (W_0[l][n][:] and phi[r][:]): each thread moves one float4 of a
    // 32-row slab per step, global -> registers -> LDS; chunk c+1 is in flight while chunk c is multiplied.
    // Named registers, no arrays: hipcc leaves a conditionally written float4 array in scratch.
    const float* W0 = a.W[0] + (size_t)l * HID * a.F;
    const int nch = a.F / BK;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, rb4, rb5, rb6;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = rb4 = rb5 = rb6 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int s_row = tid >> 3, s_c4 = tid & 7;  // 32 rows x 8 float4 per slab
    const float* a_src = W0 + (size_t)s_row * a.F + 4 * s_c4;
    const float* b_src = a.phiT + (size_t)(b0 + s_row) * a.F + 4 * s_c4;  // phi is (R, F) row-major here
    const size_t a_step = (size_t)32 * a.F;
    const size_t b_step = (size_t)a.B * a.F;  // next stencil point, same base samples
#define NSVD_LDG(p) (*reinterpret_cast<const float4*>(p))
#define NSVD_LOAD_CHUNK(c)                                                       \
    {                                                                            \
        const float* pa_ = a_src + (c) * BK;                                     \
        const float* pb_ = b_src + (c) * BK;                                     \
        ra0 = NSVD_LDG(pa_);                                                     \
        ra1 = NSVD_LDG(pa_ + a_step);                                            \
        ra2 = NSVD_LDG(pa_ + 2 * a_step);                                        \
        ra3 = NSVD_LDG(pa_ + 3 * a_step);                                        \
        rb0 = NSVD_LDG(pb_);                                                     \
        if (E > 1) rb1 = NSVD_LDG(pb_ + b_step);                                 \
        if (E > 2) rb2 = NSVD_LDG(pb_ + 2 * b_step);                             \
        if (E > 3) rb3 = NSVD_LDG(pb_ + 3 * b_step);                             \
        if (E > 4) rb4 = NSVD_LDG(pb_ + 4 * b_step);                             \
        if (E > 5) rb5 = NSVD_LDG(pb_ + 5 * b_step);                             \
        if (E > 6) rb6 = NSVD_LDG(pb_ + 6 * b_step);                             \
    }
#define NSVD_STS(p, v) (*reinterpret_cast<float4*>(p) = (v))
#define NSVD_STORE_CHUNK(buf)                                                    \
    {                                                                            \
        float* Ab_ = As + (buf) * HID * A_LD + s_row * A_LD + 4 * s_c4;          \
        float* Bb_ = Bs + (buf) * NC * A_LD + s_row * A_LD + 4 * s_c4;           \
        NSVD_STS(Ab_, ra0);                                                      \
        NSVD_STS(Ab_ + 32 * A_LD, ra1);                                          \
        NSVD_STS(Ab_ + 64 * A_LD, ra2);                                          \
        NSVD_STS(Ab_ + 96 * A_LD, ra3);                                          \
        NSVD_STS(Bb_, rb0);                                                      \
        if (E > 1) NSVD_STS(Bb_ + 32 * A_LD, rb1);                               \
        if (E > 2) NSVD_STS(Bb_ + 64 * A_LD, rb2);                               \
        if (E > 3) NSVD_STS(Bb_ + 96 * A_LD, rb3);                               \
        if (E > 4) NSVD_STS(Bb_ + 128 * A_LD, rb4);                              \
        if (E > 5) NSVD_STS(Bb_ + 160 * A_LD, rb5);                              \
        if (E > 6) NSVD_STS(Bb_ + 192 * A_LD, rb6);                              \
    }

    NSVD_LOAD_CHUNK(0);
    NSVD_STORE_CHUNK(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1 < nch);
        if (more) NSVD_LOAD_CHUNK(c + 1);
        const float* Ap = As + cur * HID * A_LD + (32 * w + li) * A_LD + 4 * hi;
        const float* Bp = Bs + cur * NC * A_LD + li * A_LD + 4 * hi;
        // two fragment sets: the reads of q-group q+1 are issued before the 4E MFMAs of q-group q
        Frag<E> f0, f1;
        load_frag<E>(f0, Ap, Bp, A_LD);
        load_frag<E>(f1, Ap + 8, Bp + 8, A_LD);
        mma_frag<E>(acc, f0);
        load_frag<E>(f0, Ap + 16, Bp + 16, A_LD);
        mma_frag<E>(acc, f1);
        load_frag<E>(f1, Ap + 24, Bp + 24, A_LD);
        mma_frag<E>(acc, f0);
        mma_frag<E>(acc, f1);
        if (more) NSVD_STORE_CHUNK(cur ^ 1);
        __syncthreads();
    }
#undef NSVD_LOAD_CHUNK
#undef NSVD_STORE_CHUNK
#undef NSVD_LDG
#undef NSVD_STS

    // ------------------------------------------------------------------ hidden layers 1 .. nh-1
    const int nh = a.nlayers - 1;
    for (int i = 0; i < nh; ++i) {
        // bias + (save centre pre-activations) + softplus, in registers
        const float* bi = a.b[i] + (size_t)l * HID + 32 * w;
        float* zs = a.zsave[i] ? a.zsave[i] + ((size_t)l * HID + 32 * w) * a.B + b0 + li : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = acc_row(r, hi);
            const float bv = bi[n];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float z = acc[e][r] + bv;
                if (e == 0 && zs) zs[(size_t)n * a.B] = z;
                acc[e][r] = nsvd_softplus(z);
            }
        }
        if (i == nh - 1) break;
        // next layer's W fragments (row 32w+li, 16 B at column 8q + 4hi) straight from L2, issued before
        // the activations move through LDS
        const float* Wn = a.W[i + 1] + ((size_t)l * HID + 32 * w + li) * HID + 4 * hi;
        float4 wf[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) wf[q] = *reinterpret_cast<const float4*>(Wn + 8 * q);
        __syncthreads();  // every wave is done reading the previous LDS contents
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
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
        const float* Hp = Hs + li * H_LD + 4 * hi;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            Frag<E> f;
            f.a = wf[q];
#pragma unroll
            for (int e = 0; e < E; ++e) f.b[e] = *reinterpret_cast<const float4*>(Hp + e * BS * H_LD + 8 * q);
            mma_frag<E>(acc, f);
        }
    }

    // ------------------------------------------------------------------ last layer 128 -> 1
    {
        const float* wl = a.W[nh] + (size_t)l * HID + 32 * w;
        float part[E];
#pragma unroll
        for (int e = 0; e < E; ++e) part[e] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float wv = wl[acc_row(r, hi)];
#pragma unroll
            for (int e = 0; e < E; ++e) part[e] = fmaf(wv, acc[e][r], part[e]);
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            part[e] += __shfl_xor(part[e], 32, 64);
            if (hi == 0) red[w * NC + e * BS + li] = part[e];
        }
    }
    __syncthreads();
    if (tid < NC) outs[tid] = (red[tid] + red[NC + tid]) + (red[2 * NC + tid] + red[3 * NC + tid]) + a.b[nh][l];
    __syncthreads();

    // ------------------------------------------------------------------ FD Hamiltonian epilogue
    if (tid < BS) {
        const int b = b0 + tid;
        float xc[NSVD_FD_MAXD];
        for (int d = 0; d < a.D; ++d) xc[d] = a.x[(size_t)b * a.D + d];
        float bv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) bv[e] = outs[e * BS + tid];
        const float s_l = a.scales ? a.scales[l] : 0.f;
        const NsvdFdOut o = nsvd_fd_point(bv, xc, a.D, a.scales != nullptr, s_l, a.prob, a.log_norm);
        const size_t idx = (size_t)b * a.L + l;
        a.f[idx] = o.f;
        a.Tf[idx] = o.Tf;
        if (a.jac) a.jac[idx] = o.jac;
        if (a.dsc) a.dsc[idx] = o.dsc;
    }
}

template <int E>
size_t fwd_lds_bytes() {
    constexpr int NC = E * BS;
    constexpr int STAGE = 2 * HID * A_LD + 2 * NC * A_LD;
    constexpr int HSZ = NC * H_LD;
    return ((STAGE > HSZ ? STAGE : HSZ) + 5 * NC) * sizeof(float);
}

template <int E>
int launch_fwd(const FwdArgs& a, hipStream_t s) {
    const size_t lds = fwd_lds_bytes<E>();
    static bool attr_done = false;  // idempotent, racing threads set the same value
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)pmlp_fused_fwd_kernel<E>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_done = true;
    }
    const int grid = (a.B / BS) * a.L;
    nsvd_prof_begin(s);
    hipLaunchKernelGGL(pmlp_fused_fwd_kernel<E>, dim3(grid), dim3(256), lds, s, a);
    nsvd_prof_end(s);
    NSVD_CHECK_LAUNCH();
    return 0;
}

// ================================================================================================
// BACKWARD, part 1 (pmlp_fused_bwd_chain_kernel): one workgroup = head l x 32 centre samples.
// Walks the layers from the output back to layer 0 with the same transposed tiles as the forward
// (rows = hidden units, columns = samples on the lanes; wave w owns rows 32w..32w+31):
//   dz_i = dh_i * sigmoid(z_i)                                   (registers)
//   dW_i[n][k] += sum_c dz_i[n][c] h_{i-1}[k][c]   (i >= 1)       64 MFMAs/wave, K = 32 samples, float atomics
//   db_i[n]    += sum_c dz_i[n][c]                                wave shuffles + atomics
//   dh_{i-1}[k][c] = sum_n W_i[n][k] dz_i[n][c]                   64 MFMAs/wave, K = 128, W_i from L2
// and leaves dz_0 (L, 128, B) in the workspace for the layer-0 weight-gradient GEMM.
// This is autograd's backward of reference mlp.py:204-221 for the centre evaluation only (the 2D
// shifted evaluations carry no gradient: nestedlora.py:108-111), plus pde/__init__.py:16 and
// boundary.py:46-53 (d/d scales).
struct ChainArgs {
    const float* df;
    const float* jac;
    const float* dsc;
    const float* W[NSVD_MAX_LAYERS];
    const float* zsave[NSVD_MAX_LAYERS];
    float* gW[NSVD_MAX_LAYERS];
    float* gb[NSVD_MAX_LAYERS];
    float* gscales;
    float* dz0;
    int nlayers, B, L;
};

constexpr int C_LD = BS + 4;  // [row][sample] images, sample contiguous

__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes that share lane>>5
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ void __launch_bounds__(256, 1) pmlp_fused_bwd_chain_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float DZ[BS * H_LD];     // [c][n]  n contiguous
    __shared__ __attribute__((aligned(16))) float DZT[HID * C_LD];   // [n][c]  c contiguous
    __shared__ __attribute__((aligned(16))) float HT[HID * C_LD];    // [k][c]  c contiguous
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nsb = a.B / BS;
    const int l = blockIdx.x / nsb;
    const int b0 = (blockIdx.x - l * nsb) * BS;
    const int nh = a.nlayers - 1;
    const int b = b0 + li;

    const float dfv = a.df[(size_t)b * a.L + l];
    const float dbase = dfv * a.jac[(size_t)b * a.L + l];
    if (w == 0) {
        const float s = half_sum(dbase);
        if (lane == 0) atomicAdd(&a.gb[nh][l], s);
        if (a.gscales) {
            const float t = half_sum(dfv * a.dsc[(size_t)b * a.L + l]);
            if (lane == 0) atomicAdd(&a.gscales[l], t);
        }
    }

    // ---- output layer (128 -> 1): dW_last, and dz of the last hidden layer
    float dz[16], hreg[16], sg[16];
    {
        const float* zp = a.zsave[nh - 1] + ((size_t)l * HID + 32 * w) * a.B + b;
        const float* wl = a.W[nh] + (size_t)l * HID + 32 * w;
        float* gwl = a.gW[nh] + (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = acc_row(r, hi);
            const float z = zp[(size_t)n * a.B];
            const float h = nsvd_softplus(z);
            const float t = half_sum(dbase * h);
            if (li == 0) atomicAdd(&gwl[n], t);
            dz[r] = wl[n] * dbase * nsvd_sigmoid(z);
            hreg[r] = h;
        }
    }

    for (int i = nh - 1; i >= 0; --i) {
        // bias gradient of layer i
        float* gbi = a.gb[i] + (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float t = half_sum(dz[r]);
            if (li == 0) atomicAdd(&gbi[acc_row(r, hi)], t);
        }
        if (i == 0) break;
        // activations feeding layer i: h_{i-1} = softplus(z_{i-1}) (this wave's 32 rows), and sigmoid for later
        {
            const float* zp = a.zsave[i - 1] + ((size_t)l * HID + 32 * w) * a.B + b;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = zp[(size_t)acc_row(r, hi) * a.B];
                hreg[r] = nsvd_softplus(z);
                sg[r] = nsvd_sigmoid(z);
            }
        }
        __syncthreads();  // previous round's LDS reads are done
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(&DZ[li * H_LD + 32 * w + 8 * g + 4 * hi]) =
                make_float4(dz[4 * g], dz[4 * g + 1], dz[4 * g + 2], dz[4 * g + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * w + acc_row(r, hi);
            DZT[n * C_LD + li] = dz[r];
            HT[n * C_LD + li] = hreg[r];
        }
        __syncthreads();

        // ---- weight gradient: rows n = 32w.., all 128 k, K = 32 samples
        {
            f32x16 acc[4];
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[kb][r] = 0.f;
            const float* Ap = DZT + (32 * w + li) * C_LD + 4 * hi;
            const float* Bp = HT + li * C_LD + 4 * hi;
#pragma unroll
            for (int q = 0; q < BS / 8; ++q) {
                Frag<4> f;
                load_frag<4>(f, Ap + 8 * q, Bp + 8 * q, C_LD);
                mma_frag<4>(acc, f);
            }
            float* gw = a.gW[i] + ((size_t)l * HID + 32 * w) * HID + li;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) atomicAdd(&gw[(size_t)acc_row(r, hi) * HID + 32 * kb], acc[kb][r]);
        }
        // ---- data gradient: rows k = 32w.., K = 128 hidden units of layer i
        {
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
            for (int r = 0; r < 16; ++r) dz[r] = acc1[0][r] * sg[r];
        }
    }
    // dz_0 -> workspace (L, 128, B)
    float* o = a.dz0 + ((size_t)l * HID + 32 * w) * a.B + b;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)acc_row(r, hi) * a.B] = dz[r];
}

// ================================================================================================
// BACKWARD, part 2 (pmlp_fused_wgrad0_kernel): dW_0[l][n][k] = sum_b dz_0[l][n][b] phi^T[k][b] over the
// B centre samples: per head a 128 x F x B GEMM with both operands b-contiguous. 128 x 128 output tile
// per workgroup (F/128 * L workgroups), 4 waves as 2 x 2 of 64 x 64 (2 x 2 MFMA tiles each), K streamed
// in 32-sample chunks through padded LDS tiles, register-staged and double buffered like the forward.
struct Wgrad0Args {
    const float* dz0;    // (L, 128, B)
    const float* phiTc;  // (F, B)
    float* gW0;          // (L, 128, F)
    int B, L, F;
    int xcd_remap;
};

__global__ void __launch_bounds__(256, 1) pmlp_fused_wgrad0_kernel(Wgrad0Args a) {
    __shared__ __attribute__((aligned(16))) float As[2 * HID * A_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2 * HID * A_LD];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nkt = a.F / HID;
    int unit = blockIdx.x;
    if (a.xcd_remap) {
        const int per = gridDim.x >> 3;
        unit = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    }
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
    const float* a_src = a.dz0 + ((size_t)l * HID + s_row) * a.B + 4 * s_c4;
    const float* b_src = a.phiTc + (size_t)(kf0 + s_row) * a.B + 4 * s_c4;
    const size_t step = (size_t)32 * a.B;
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define W0_LOAD(c)                                                                    \
    {                                                                                 \
        const float* pa_ = a_src + (c) * BK;                                          \
        const float* pb_ = b_src + (c) * BK;                                          \
        ra0 = *reinterpret_cast<const float4*>(pa_);                                  \
        ra1 = *reinterpret_cast<const float4*>(pa_ + step);                           \
        ra2 = *reinterpret_cast<const float4*>(pa_ + 2 * step);                       \
        ra3 = *reinterpret_cast<const float4*>(pa_ + 3 * step);                       \
        rb0 = *reinterpret_cast<const float4*>(pb_);                                  \
        rb1 = *reinterpret_cast<const float4*>(pb_ + step);                           \
        rb2 = *reinterpret_cast<const float4*>(pb_ + 2 * step);                       \
        rb3 = *reinterpret_cast<const float4*>(pb_ + 3 * step);                       \
    }
#define W0_STORE(buf)                                                                 \
    {                                                                                 \
        float* Ab_ = As + (buf) * HID * A_LD + s_row * A_LD + 4 * s_c4;               \
        float* Bb_ = Bs + (buf) * HID * A_LD + s_row * A_LD + 4 * s_c4;               \
        *reinterpret_cast<float4*>(Ab_) = ra0;                                        \
        *reinterpret_cast<float4*>(Ab_ + 32 * A_LD) = ra1;                            \
        *reinterpret_cast<float4*>(Ab_ + 64 * A_LD) = ra2;                            \
        *reinterpret_cast<float4*>(Ab_ + 96 * A_LD) = ra3;                            \
        *reinterpret_cast<float4*>(Bb_) = rb0;                                        \
        *reinterpret_cast<float4*>(Bb_ + 32 * A_LD) = rb1;                            \
        *reinterpret_cast<float4*>(Bb_ + 64 * A_LD) = rb2;                            \
        *reinterpret_cast<float4*>(Bb_ + 96 * A_LD) = rb3;                            \
    }
    const int nch = a.B / BK;
    W0_LOAD(0);
    W0_STORE(0);
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
        const int cur = c & 1;
        const bool more = (c + 1 < nch);
        if (more) W0_LOAD(c + 1);
        const float* Ap = As + cur * HID * A_LD + (64 * wm + li) * A_LD + 4 * hi;
        const float* Bp = Bs + cur * HID * A_LD + (64 * wn + li) * A_LD + 4 * hi;
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
            const float4 a0 = *reinterpret_cast<const float4*>(Ap + 8 * q);
            const float4 a1 = *reinterpret_cast<const float4*>(Ap + 32 * A_LD + 8 * q);
            const float4 b0v = *reinterpret_cast<const float4*>(Bp + 8 * q);
            const float4 b1v = *reinterpret_cast<const float4*>(Bp + 32 * A_LD + 8 * q);
#define W0_MMA(X)                                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, b0v.X, acc[0][0], 0, 0, 0);                 \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, b1v.X, acc[0][1], 0, 0, 0);                 \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, b0v.X, acc[1][0], 0, 0, 0);                 \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, b1v.X, acc[1][1], 0, 0, 0);
            W0_MMA(x) W0_MMA(y) W0_MMA(z) W0_MMA(w)
#undef W0_MMA
        }
        if (more) W0_STORE(cur ^ 1);
        __syncthreads();
    }
#undef W0_LOAD
#undef W0_STORE
    float* o = a.gW0 + ((size_t)l * HID + 64 * wm) * a.F + kf0 + 64 * wn + li;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)(32 * i + acc_row(r, hi)) * a.F + 32 * j] = acc[i][j][r];
}

// zero the gradient tensors that the chain kernel accumulates into with atomics
struct ZeroArgs {
    float* p[2 * NSVD_MAX_LAYERS + 1];
    unsigned n[2 * NSVD_MAX_LAYERS + 1];
    int count;
};
__global__ void __launch_bounds__(256) zero_many_kernel(ZeroArgs z) {
    for (int t = 0; t < z.count; ++t) {
        float* p = z.p[t];
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < z.n[t]; i += gridDim.x * blockDim.x) p[i] = 0.f;
    }
}

struct FusedWs {
    float* phi;                       // (R, F) sample-major Fourier features of every stencil row
    float* phiTc;                     // (F, B) feature-major copy of the centre rows (weight gradient)
    float* zsave[NSVD_MAX_LAYERS];    // (L, 128, B) per hidden layer
    float* jac;                       // (B, L)
    float* dsc;                       // (B, L)
    float* dz0;                       // (L, 128, B)
    size_t bytes;
};

FusedWs carve_fused(const nsvd_model_desc& d, int B, void* base) {
    FusedWs w;
    memset(&w, 0, sizeof(w));
    const size_t E = 1 + 2 * (size_t)d.D, R = E * B, F = 2 * (size_t)d.m;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float* q = (float*)(p + off);
        off += nsvd_align(nfloats * sizeof(float));
        return q;
    };
    w.phi = take(F * R);
    w.phiTc = take(F * B);
    for (int i = 0; i < d.nlayers - 1; ++i) w.zsave[i] = take((size_t)d.L * HID * B);
    w.jac = take((size_t)B * d.L);
    w.dsc = take((size_t)B * d.L);
    w.dz0 = take((size_t)d.L * HID * B);
    w.bytes = off;
    return w;
}

}  // namespace

bool nsvd_fused_supported(const nsvd_model_desc& d, int B) {
    if (d.D < 1 || d.D > 3) return false;
    if (d.nlayers < 2) return false;
    for (int i = 0; i < d.nlayers - 1; ++i)
        if (d.dims[i] != HID) return false;
    if (B % BS != 0) return false;
    if ((2 * d.m) % HID != 0) return false;  // layer-0 weight gradient uses 128-wide feature tiles
    return true;
}

size_t nsvd_fused_workspace_bytes(const nsvd_model_desc& d, int B) { return carve_fused(d, B, nullptr).bytes; }

int nsvd_fused_forward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                       int B, float* f, float* Tf, void* ws, int save, hipStream_t s) {
    const FusedWs w = carve_fused(d, B, ws);
    const int E = 1 + 2 * d.D, R = E * B, F = 2 * d.m;
    int rc = nsvd_fourier_rows(x, p.fourier_B, w.phi, B, d.D, d.m, prob.eps, E, s);
    if (rc) return rc;
    if (save) {
        rc = nsvd_fourier_features(x, p.fourier_B, w.phiTc, B, d.D, d.m, prob.eps, 1, B, s);
        if (rc) return rc;
    }
    FwdArgs a;
    memset(&a, 0, sizeof(a));
    a.phiT = w.phi;
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
    const int grid = (B / BS) * d.L;
    a.xcd_remap = (grid % 8 == 0) ? 1 : 0;
    switch (E) {
        case 3: return launch_fwd<3>(a, s);
        case 5: return launch_fwd<5>(a, s);
        case 7: return launch_fwd<7>(a, s);
    }
    return NSVD_EUNSUPPORTED;
}

int nsvd_fused_backward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                        int B, const float* df, const nsvd_params& g, void* ws, hipStream_t s) {
    (void)prob;
    (void)x;
    const FusedWs w = carve_fused(d, B, ws);
    const int F = 2 * d.m, nh = d.nlayers - 1;
    ZeroArgs z;
    memset(&z, 0, sizeof(z));
    int c = 0;
    for (int i = 0; i < d.nlayers; ++i) {
        if (i > 0) {
            z.p[c] = g.W[i];
            z.n[c++] = (unsigned)((size_t)d.L * d.dims[i] * d.dims[i - 1]);
        }
        z.p[c] = g.b[i];
        z.n[c++] = (unsigned)((size_t)d.L * d.dims[i]);
    }
    if (d.has_exp_mask) {
        z.p[c] = g.scales;
        z.n[c++] = (unsigned)d.L;
    }
    z.count = c;
    hipLaunchKernelGGL(zero_many_kernel, dim3(256), dim3(256), 0, s, z);
    NSVD_CHECK_LAUNCH();

    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.df = df;
    a.jac = w.jac;
    a.dsc = d.has_exp_mask ? w.dsc : nullptr;
    for (int i = 0; i < d.nlayers; ++i) {
        a.W[i] = p.W[i];
        a.zsave[i] = (i < nh) ? w.zsave[i] : nullptr;
        a.gW[i] = g.W[i];
        a.gb[i] = g.b[i];
    }
    a.gscales = d.has_exp_mask ? g.scales : nullptr;
    a.dz0 = w.dz0;
    a.nlayers = d.nlayers; a.B = B; a.L = d.L;
    hipLaunchKernelGGL(pmlp_fused_bwd_chain_kernel, dim3((B / BS) * d.L), dim3(256), 0, s, a);
    NSVD_CHECK_LAUNCH();

    Wgrad0Args wa;
    wa.dz0 = w.dz0;
    wa.phiTc = w.phiTc;
    wa.gW0 = g.W[0];
    wa.B = B; wa.L = d.L; wa.F = F;
    const int grid = (F / HID) * d.L;
    wa.xcd_remap = (grid % 8 == 0) ? 1 : 0;
    hipLaunchKernelGGL(pmlp_fused_wgrad0_kernel, dim3(grid), dim3(256), 0, s, wa);
    NSVD_CHECK_LAUNCH();
    return 0;
}
