// NestedLoRA EVD loss: moments, nesting masks, loss scalar and d loss / d f.
//   reference: compute_lambda                       methods/nestedlora.py:10-11
//              get_joint/sequential_nesting_masks   methods/nestedlora.py:40-54
//              compute_loss_metric                  methods/nestedlora.py:57-64
//              NestedLoRALossFunctionEVD.forward    methods/nestedlora.py:70-94
//              NestedLoRALossFunctionEVD.backward   methods/nestedlora.py:98-111
//              f1, f2 = torch.chunk(f, 2)           methods/nestedlora.py:263
// Three small launches: per-chunk partial moments (deterministic: no float atomics) -> fixed-order
// reduction into the (2 L^2 + 1)-float exchange payload -> loss + gradient.  Rows are staged in LDS;
// the mask is generated in registers for the sequential / joint(step 1) nestings.
#include "nsvd_kernels.h"
#include "evd_math.h"

namespace {

constexpr int CH = NSVD_EVD_CH;  // rows per chunk
constexpr int MAXL = 128;        // LDS budget: CH * MAXL floats

__device__ __forceinline__ float mask_v(int kind, const float* v, int l, int L) { return nsvd_mask_v(kind, v, l, L); }
__device__ __forceinline__ float mask_M(int kind, const float* M, int l, int m, int L) {
    return nsvd_mask_M(kind, M, l, m, L);
}
typedef NsvdEvdChunking Chunking;
__host__ __device__ inline Chunking chunking(int B) { return nsvd_evd_chunking(B); }
__device__ __forceinline__ void chunk_rows(const Chunking& c, int chunk, int& r0, int& r1) {
    if (chunk < c.n1) { r0 = chunk * CH; r1 = min(r0 + CH, c.B1); }
    else { r0 = c.B1 + (chunk - c.n1) * CH; r1 = min(r0 + CH, c.B1 + c.B2); }
}

// part[i][j] = sum_r fs[r][i] fs[r][j] over the chunk's rows, 16 outputs per thread at a time: every output is still
// summed over r in order (the same bits as one output at a time), but the 16 accumulator chains are independent - one
// chain per thread was a dependent LDS-read + FMA sequence of 16 x nr steps (30 us at L = 64, B = 8192: 128 workgroups
// of 64 rows), now the LDS reads of a row are in flight together
// NQ outputs per thread and pass (4 covers L <= 32 in one pass without idle accumulators); JC: 256 is a multiple of L, so
// a thread's column j = tid % L is the same for all of its outputs - one read of the row's element instead of NQ
template <int NQ, bool JC>
__device__ __forceinline__ void chunk_gram_t(const float* fs, int nr, int L, float* __restrict__ out, int o0, int o1) {
    const int LL = L * L;
    for (int base = o0; base < o1; base += 256 * NQ) {
        float acc[NQ];
        int ii[NQ], jj[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int o = min(base + (int)threadIdx.x + 256 * q, LL - 1);
            ii[q] = o / L;
            jj[q] = o - ii[q] * L;
            acc[q] = 0.f;
        }
        for (int r = 0; r < nr; ++r) {
            const float* row = fs + r * L;
            const float rj = row[jj[0]];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = fmaf(row[ii[q]], JC ? rj : row[jj[q]], acc[q]);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int o = base + (int)threadIdx.x + 256 * q;
            if (o < o1) out[o] = acc[q];
        }
    }
}
// outputs [o0, o1) of the chunk's L x L block (o0 a multiple of 256; the whole block: 0, L * L)
__device__ __forceinline__ void chunk_gram(const float* fs, int nr, int L, float* __restrict__ out, int o0, int o1) {
    const bool jc = 256 % L == 0 && L * L >= 256;  // (every output of a thread then exists or is the clamped last one)
    if (o1 - o0 <= 512) {
        if (jc) chunk_gram_t<2, true>(fs, nr, L, out, o0, o1);
        else chunk_gram_t<2, false>(fs, nr, L, out, o0, o1);
    } else if (o1 - o0 <= 1024) {
        if (jc) chunk_gram_t<4, true>(fs, nr, L, out, o0, o1);
        else chunk_gram_t<4, false>(fs, nr, L, out, o0, o1);
    } else {
        if (jc) chunk_gram_t<16, true>(fs, nr, L, out, o0, o1);
        else chunk_gram_t<16, false>(fs, nr, L, out, o0, o1);
    }
}
// a chunk's block is cut over this many workgroups (gridDim.y): 64-row chunks give B / 64 workgroups - 128 at
// configs[3] - each walking L^2 / 256 outputs per thread through LDS reads; cut four ways the kernel is 4 x as parallel (24.9 -> 17.0 us; eight ways: the same - what is left is the staging)
// (every output is summed by one thread over the same rows in the same order: the same bits)
inline int evd_partial_cuts(int B, int L) {
    const int nch = chunking(B).n1 + chunking(B).n2;
    int cuts = 1;
    while (cuts < 8 && nch * cuts < 512 && (L * L) % (256 * 2 * cuts) == 0 && (L * L) / (2 * cuts) >= 1024) cuts *= 2;
    return cuts;
}

__global__ void __launch_bounds__(256) evd_partial_kernel(const float* __restrict__ f, const float* __restrict__ Tf,
                                                          int B, int L, int kind, const float* __restrict__ v,
                                                          float* __restrict__ part, float* __restrict__ part_op) {
    extern __shared__ __attribute__((aligned(16))) float fs[];  // [CH][L]
    __shared__ float red[4];
    const Chunking c = chunking(B);
    int r0, r1;
    chunk_rows(c, blockIdx.x, r0, r1);
    const int nr = r1 - r0;
    float op = 0.f;
    for (int i = threadIdx.x; i < nr * L; i += 256) {
        const float fv = f[(size_t)r0 * L + i];
        fs[i] = fv;
        const int l = i % L;
        op = fmaf(mask_v(kind, v, l, L) * fv, Tf[(size_t)r0 * L + i], op);
    }
    __syncthreads();
    {
        const int per = L * L / (int)gridDim.y;  // (gridDim.y > 1: a multiple of 256, evd_partial_cuts)
        const int o0 = per * (int)blockIdx.y;
        chunk_gram(fs, nr, L, part + (size_t)blockIdx.x * L * L, o0, blockIdx.y + 1 == gridDim.y ? L * L : o0 + per);
    }
    if (blockIdx.y != 0) return;
    op = nsvd_wave_sum(op);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = op;
    __syncthreads();
    if (threadIdx.x == 0) part_op[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Head-sharded runs: the ranks' packed [f | Tf] blocks as the all-gather leaves them -> the (B, L) arrays every
// consumer reads, and (part != null) the per-chunk partial moments of evd_partial_kernel in the same pass - the same
// loop, the same accumulation order: bit-identical to a permuting copy followed by that kernel.
// Rank w owns n_w = L / W + (w < L % W) consecutive heads (the first L % W ranks one more: include/nsvd.h); its block
// starts at gath + w * 2 B Lb (Lb = ceil(L / W): all-gather blocks are equally long) and holds f (B, n_w) then
// Tf (B, n_w), packed - with L % W == 0 that is the plain (W, 2, B, L / W) array.
__global__ void __launch_bounds__(256) evd_gather_heads_kernel(const float* __restrict__ gath, int W, int B, int L,
                                                               int Lb, int kind, const float* __restrict__ v,
                                                               float* __restrict__ f, float* __restrict__ Tf,
                                                               float* __restrict__ part, float* __restrict__ part_op) {
    extern __shared__ __attribute__((aligned(16))) float fs[];  // [CH][L]
    __shared__ float red[4];
    const int base = L / W, rem = L - base * W, big = rem * (base + 1);
    const Chunking c = chunking(B);
    int r0, r1;
    chunk_rows(c, blockIdx.x, r0, r1);
    const int nr = r1 - r0;
    float op = 0.f;
    for (int i = threadIdx.x; i < nr * L; i += 256) {
        const int r = i / L, l = i - r * L;
        int w, ll, n;
        if (l < big) {
            n = base + 1; w = l / n; ll = l - w * n;
        } else {
            n = base; w = rem + (l - big) / n; ll = (l - big) - (w - rem) * n;
        }
        const size_t src = (size_t)w * 2 * B * Lb + (size_t)(r0 + r) * n + ll;
        const float fv = gath[src];
        const float tv = gath[src + (size_t)B * n];
        f[(size_t)r0 * L + i] = fv;
        Tf[(size_t)r0 * L + i] = tv;
        if (part) {
            fs[i] = fv;
            op = fmaf(mask_v(kind, v, l, L) * fv, tv, op);
        }
    }
    if (!part) return;
    __syncthreads();
    {
        const int per = L * L / (int)gridDim.y;  // (gridDim.y > 1: a multiple of 256, evd_partial_cuts)
        const int o0 = per * (int)blockIdx.y;
        chunk_gram(fs, nr, L, part + (size_t)blockIdx.x * L * L, o0, blockIdx.y + 1 == gridDim.y ? L * L : o0 + per);
    }
    if (blockIdx.y != 0) return;
    op = nsvd_wave_sum(op);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = op;
    __syncthreads();
    if (threadIdx.x == 0) part_op[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) evd_reduce_kernel(const float* __restrict__ part,
                                                         const float* __restrict__ part_op, int B, int L,
                                                         float* __restrict__ moments) {
    const Chunking c = chunking(B);
    const int LL = L * L;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < LL) {
        // chunk order, as everywhere these partials are summed (nsvd_evd_lam): the same bits; the loads of eight chunks
        // are issued before the first is added (17 workgroups walk up to 128 chunks each: the kernel is one dependent
        // load chain per thread otherwise - 36 us at L = 64, B = 8192)
        auto sum_chunks = [&](const float* p0, int n) {
            float s = 0.f;
            int k = 0;
            for (; k + 8 <= n; k += 8) {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = p0[(size_t)(k + j) * LL];
#pragma unroll
                for (int j = 0; j < 8; ++j) s += t[j];
            }
            for (; k < n; ++k) s += p0[(size_t)k * LL];
            return s;
        };
        const float s1 = sum_chunks(part + i, c.n1);
        const float s2 = sum_chunks(part + (size_t)c.n1 * LL + i, c.n2);
        moments[i] = s1 / (float)c.B1;
        moments[LL + i] = s2 / (float)c.B2;  // B2 == 0 (B == 1) gives nan, like the reference
    } else if (i == LL) {
        float s = 0.f;
        for (int k = 0; k < c.n1 + c.n2; ++k) s += part_op[k];
        moments[2 * LL] = s / (float)B;
    }
}

__global__ void __launch_bounds__(256) evd_loss_grad_kernel(const float* __restrict__ f, const float* __restrict__ Tf,
                                                            int B, int L, int kind, const float* __restrict__ v,
                                                            const float* __restrict__ M,
                                                            const float* __restrict__ moments, float grad_scale,
                                                            float* __restrict__ loss, float* __restrict__ df) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // ML[L][L] then fs[CH][L]
    __shared__ float red[4];
    float* ML = sm;
    float* fs = sm + L * L;
    const Chunking c = chunking(B);
    const int LL = L * L;
    int r0, r1;
    chunk_rows(c, blockIdx.x, r0, r1);
    const int nr = r1 - r0;
    const bool first_half = blockIdx.x < c.n1;
    const float* lam_other = moments + (first_half ? LL : 0);
    for (int o = threadIdx.x; o < LL; o += 256) {
        const int l = o / L, m = o - l * L;
        ML[o] = mask_M(kind, M, l, m, L) * lam_other[o];
    }
    if (blockIdx.x == 0 && loss) {
        float s = 0.f;
        for (int o = threadIdx.x; o < LL; o += 256) {
            const int l = o / L, m = o - l * L;
            s = fmaf(mask_M(kind, M, l, m, L) * moments[o], moments[LL + o], s);
        }
        s = nsvd_wave_sum(s);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    }
    if (df) {
        for (int i = threadIdx.x; i < nr * L; i += 256) fs[i] = f[(size_t)r0 * L + i];
    }
    __syncthreads();
    if (blockIdx.x == 0 && loss && threadIdx.x == 0) {
        const float metric = red[0] + red[1] + red[2] + red[3];
        const float oper = -2.f * moments[2 * LL];
        loss[0] = oper + metric;
        loss[1] = oper;
        loss[2] = metric;
    }
    if (!df) return;
    const float cop = -4.f / (float)B;
    const float cm = 2.f / (float)(first_half ? c.B1 : c.B2);
    for (int i = threadIdx.x; i < nr * L; i += 256) {
        const int r = i / L, m = i - r * L;
        float s = 0.f;
        for (int l = 0; l < L; ++l) s = fmaf(fs[r * L + l], ML[l * L + m], s);
        const float g = cop * mask_v(kind, v, m, L) * Tf[(size_t)r0 * L + i] + cm * s;
        df[(size_t)r0 * L + i] = grad_scale * g;
    }
}

// Single-workgroup variant for small problems (B*L <= FUSED_MAX floats per operand in LDS): moments,
// loss and (optionally) the gradient in ONE launch - the three-launch pipeline above costs ~5 us per
// launch for ~0.5 us of work at B = 512, L = 16. STAGE: 1 = moments only (data-parallel stage 1),
// 3 = moments + loss + gradient (single GPU).
constexpr int FUSED_MAX = 16384;
constexpr int FUSED_THREADS = 1024;

template <int STAGE>
__global__ void __launch_bounds__(FUSED_THREADS) evd_fused_kernel(const float* __restrict__ f,
                                                                  const float* __restrict__ Tf, int B, int L, int kind,
                                                                  const float* __restrict__ v,
                                                                  const float* __restrict__ M, float grad_scale,
                                                                  float* __restrict__ moments,
                                                                  float* __restrict__ loss, float* __restrict__ df) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* fs = sm;               // [B][L]
    float* lam = sm + B * L;      // [2][L][L]
    float* red = lam + 2 * L * L; // [32]
    const int tid = threadIdx.x, nt = FUSED_THREADS;
    const int B1 = (B + 1) / 2, B2 = B - B1, LL = L * L;
    float op = 0.f;
    const int n = B * L;
    if ((n & 3) == 0) {
        // 16-byte loads, 4 independent pairs in flight per thread: a scalar loop here is a chain of
        // exposed HBM round trips (8 x ~2 us at B*L = 8192 on an otherwise idle chip)
        const float4* f4 = reinterpret_cast<const float4*>(f);
        const float4* T4 = reinterpret_cast<const float4*>(Tf);
        const int n4 = n >> 2;
#pragma unroll 4
        for (int i = tid; i < n4; i += nt) {
            const float4 a = f4[i], t = T4[i];
            *reinterpret_cast<float4*>(fs + 4 * i) = a;
            const int l0 = (4 * i) % L;
            op = fmaf(mask_v(kind, v, l0, L) * a.x, t.x, op);
            op = fmaf(mask_v(kind, v, (l0 + 1) % L, L) * a.y, t.y, op);
            op = fmaf(mask_v(kind, v, (l0 + 2) % L, L) * a.z, t.z, op);
            op = fmaf(mask_v(kind, v, (l0 + 3) % L, L) * a.w, t.w, op);
        }
    } else {
        for (int i = tid; i < n; i += nt) {
            const float fv = f[i];
            fs[i] = fv;
            op = fmaf(mask_v(kind, v, i % L, L) * fv, Tf[i], op);
        }
    }
    op = nsvd_wave_sum(op);
    if ((tid & 63) == 0) red[tid >> 6] = op;
    __syncthreads();
    // each (half, i, j) moment: split its B/2 rows over `parts` threads when there are spare threads
    const int outs = 2 * LL;
    int parts = 1;  // largest power of two <= min(16, threads per output)
    while (parts * 2 <= 16 && parts * 2 * outs <= nt) parts *= 2;
    for (int o = tid / parts; o < outs; o += nt / parts) {
        const int h = o / LL, ij = o - h * LL, i = ij / L, j = ij - i * L;
        const int r0 = h ? B1 : 0, nr = h ? B2 : B1;
        const int part = tid % parts;
        float s = 0.f;
        for (int r = part; r < nr; r += parts) s = fmaf(fs[(r0 + r) * L + i], fs[(r0 + r) * L + j], s);
        for (int off = 1; off < parts; off <<= 1) s += __shfl_xor(s, off, 64);  // parts is a power of two <= 16
        if (part == 0) lam[o] = s / (float)nr;
    }
    __syncthreads();
    float opm = 0.f;
    if (tid == 0) {
        for (int w = 0; w < FUSED_THREADS / 64; ++w) opm += red[w];
        opm /= (float)B;
    }
    for (int o = tid; o < outs; o += nt) moments[o] = lam[o];
    if (tid == 0) moments[2 * LL] = opm;
    if (STAGE == 1) return;
    // loss
    float s = 0.f;
    for (int o = tid; o < LL; o += nt) s = fmaf(mask_M(kind, M, o / L, o % L, L) * lam[o], lam[LL + o], s);
    s = nsvd_wave_sum(s);
    __syncthreads();
    if ((tid & 63) == 0) red[16 + (tid >> 6)] = s;
    __syncthreads();
    if (tid == 0) {
        float metric = 0.f;
        for (int w = 0; w < FUSED_THREADS / 64; ++w) metric += red[16 + w];
        const float oper = -2.f * opm;
        loss[0] = oper + metric;
        loss[1] = oper;
        loss[2] = metric;
    }
    if (!df) return;
    // masked moments in place: lam[h] <- M * lam[h]
    for (int o = tid; o < outs; o += nt) {
        const int ij = o % LL;
        lam[o] *= mask_M(kind, M, ij / L, ij % L, L);
    }
    __syncthreads();
    const float cop = -4.f / (float)B;
#pragma unroll 4
    for (int i = tid; i < B * L; i += nt) {
        const int r = i / L, m = i - r * L;
        const bool first = r < B1;
        const float* ML = lam + (first ? LL : 0);  // the OTHER half's moments
        const float tfv = Tf[i];
        float acc = 0.f;
        for (int l = 0; l < L; ++l) acc = fmaf(fs[r * L + l], ML[l * L + m], acc);
        const float g = cop * mask_v(kind, v, m, L) * tfv + (2.f / (float)(first ? B1 : B2)) * acc;
        df[i] = grad_scale * g;
    }
}

bool fused_ok(int B, int L) { return (size_t)B * L <= FUSED_MAX && L <= 64 && B >= 2; }

size_t fused_lds(int B, int L) { return ((size_t)B * L + 2 * (size_t)L * L + 32) * sizeof(float); }

template <int STAGE>
int launch_fused(const float* f, const float* Tf, int B, int L, int kind, const float* v, const float* M,
                 float grad_scale, float* moments, float* loss, float* df, hipStream_t s) {
    const size_t lds = fused_lds(B, L);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)evd_fused_kernel<STAGE>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
    }
    hipLaunchKernelGGL(evd_fused_kernel<STAGE>, dim3(1), dim3(FUSED_THREADS), lds, s, f, Tf, B, L, kind, v, M,
                       grad_scale, moments, loss, df);
    NSVD_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int nsvd_evd_loss_fused(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                                   const float* M, float grad_scale, float* moments, float* loss, float* df,
                                   void* scratch, void* stream) {
    if (!f || !Tf || !moments || !loss || B <= 0 || L <= 0) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    if (mask_kind == NSVD_MASK_CUSTOM && (!v || !M)) return NSVD_EINVAL;
    if (fused_ok(B, L))
        return launch_fused<3>(f, Tf, B, L, mask_kind, v, M, grad_scale, moments, loss, df, (hipStream_t)stream);
    int rc = nsvd_evd_moments(f, Tf, B, L, mask_kind, v, moments, scratch, stream);
    if (rc) return rc;
    return nsvd_evd_loss_grad(f, Tf, B, L, mask_kind, v, M, moments, grad_scale, loss, df, stream);
}

extern "C" size_t nsvd_evd_scratch_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return 0;
    const Chunking c = chunking(B);
    return nsvd_align((size_t)(c.n1 + c.n2) * ((size_t)L * L + 1) * sizeof(float));
}

int nsvd_evd_reduce_partials(const void* scratch, int B, int L, float* moments, hipStream_t s) {
    const Chunking c = chunking(B);
    const float* part = (const float*)scratch;
    const float* part_op = part + (size_t)(c.n1 + c.n2) * L * L;
    hipLaunchKernelGGL(evd_reduce_kernel, dim3(nsvd_cdiv(L * L + 1, 256)), dim3(256), 0, s, part, part_op, B, L,
                       moments);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_evd_partial(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                                void* scratch, void* stream) {
    if (!f || !Tf || !scratch || B <= 0 || L <= 0) return NSVD_EINVAL;
    if (L > MAXL) return NSVD_EUNSUPPORTED;
    if (mask_kind == NSVD_MASK_CUSTOM && !v) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    const Chunking c = chunking(B);
    const int nch = c.n1 + c.n2;
    float* part = (float*)scratch;
    float* part_op = part + (size_t)nch * L * L;
    hipLaunchKernelGGL(evd_partial_kernel, dim3(nch, evd_partial_cuts(B, L)), dim3(256), (size_t)CH * L * sizeof(float), (hipStream_t)stream,
                       f, Tf, B, L, mask_kind, v, part, part_op);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_evd_gather_heads(const float* gathered, int world, int B, int L_local, int mask_kind,
                                     const float* v, float* f, float* Tf, void* scratch, void* stream) {
    if (world <= 0 || L_local <= 0) return NSVD_EINVAL;
    return nsvd_evd_gather_head_blocks(gathered, world, B, world * L_local, mask_kind, v, f, Tf, scratch, stream);
}

extern "C" int nsvd_evd_gather_head_blocks(const float* gathered, int world, int B, int L, int mask_kind,
                                           const float* v, float* f, float* Tf, void* scratch, void* stream) {
    if (!gathered || !f || !Tf || world <= 0 || B <= 0 || L < world) return NSVD_EINVAL;
    const int L_block = (L + world - 1) / world;
    if (L > MAXL) return NSVD_EUNSUPPORTED;
    if (scratch && mask_kind == NSVD_MASK_CUSTOM && !v) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    const Chunking c = chunking(B);
    const int nch = c.n1 + c.n2;
    float* part = (float*)scratch;
    float* part_op = part ? part + (size_t)nch * L * L : nullptr;
    hipLaunchKernelGGL(evd_gather_heads_kernel, dim3(nch), dim3(256), (size_t)CH * L * sizeof(float),
                       (hipStream_t)stream, gathered, world, B, L, L_block, mask_kind, v, f, Tf, part, part_op);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_evd_moments(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                                float* moments, void* scratch, void* stream) {
    if (!f || !Tf || !moments || !scratch || B <= 0 || L <= 0) return NSVD_EINVAL;
    if (L > MAXL) return NSVD_EUNSUPPORTED;
    if (mask_kind == NSVD_MASK_CUSTOM && !v) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    if (fused_ok(B, L))
        return launch_fused<1>(f, Tf, B, L, mask_kind, v, nullptr, 1.f, moments, nullptr, nullptr,
                               (hipStream_t)stream);
    const Chunking c = chunking(B);
    const int nch = c.n1 + c.n2;
    float* part = (float*)scratch;
    float* part_op = part + (size_t)nch * L * L;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(evd_partial_kernel, dim3(nch, evd_partial_cuts(B, L)), dim3(256), (size_t)CH * L * sizeof(float), s, f, Tf, B, L,
                       mask_kind, v, part, part_op);
    NSVD_CHECK_LAUNCH();
    hipLaunchKernelGGL(evd_reduce_kernel, dim3(nsvd_cdiv(L * L + 1, 256)), dim3(256), 0, s, part, part_op, B, L,
                       moments);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_evd_loss_grad(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                                  const float* M, const float* moments, float grad_scale, float* loss, float* df,
                                  void* stream) {
    if (!f || !Tf || !moments || B <= 0 || L <= 0) return NSVD_EINVAL;
    if (L > MAXL) return NSVD_EUNSUPPORTED;
    if (mask_kind == NSVD_MASK_CUSTOM && (!v || !M)) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    const Chunking c = chunking(B);
    const int nch = df ? c.n1 + c.n2 : 1;
    const size_t lds = ((size_t)L * L + (size_t)CH * L) * sizeof(float);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)evd_loss_grad_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
    }
    hipLaunchKernelGGL(evd_loss_grad_kernel, dim3(nch), dim3(256), lds, (hipStream_t)stream, f, Tf, B, L, mask_kind,
                       v, M, moments, grad_scale, loss, df);
    NSVD_CHECK_LAUNCH();
    return 0;
}
