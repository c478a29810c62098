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

namespace {

constexpr int CH = 64;       // rows per chunk
constexpr int MAXL = 128;    // LDS budget: CH * MAXL floats

__device__ __forceinline__ float mask_v(int kind, const float* v, int l, int L) {
    if (kind == NSVD_MASK_SEQUENTIAL) return 1.f;
    if (kind == NSVD_MASK_JOINT) return (float)(L - l) / (float)L;
    return v[l];
}
__device__ __forceinline__ float mask_M(int kind, const float* M, int l, int m, int L) {
    if (kind == NSVD_MASK_SEQUENTIAL) return l <= m ? 1.f : 0.f;
    if (kind == NSVD_MASK_JOINT) return (float)(L - max(l, m)) / (float)L;  // min(v_l, v_m)
    return M[l * L + m];
}

struct Chunking {
    int B1, B2, n1, n2;
};
__host__ __device__ inline Chunking chunking(int B) {
    Chunking c;
    c.B1 = (B + 1) / 2;  // torch.chunk: first half gets the ceil
    c.B2 = B - c.B1;
    c.n1 = (c.B1 + CH - 1) / CH;
    c.n2 = (c.B2 + CH - 1) / CH;
    return c;
}
__device__ __forceinline__ void chunk_rows(const Chunking& c, int chunk, int& r0, int& r1) {
    if (chunk < c.n1) { r0 = chunk * CH; r1 = min(r0 + CH, c.B1); }
    else { r0 = c.B1 + (chunk - c.n1) * CH; r1 = min(r0 + CH, c.B1 + c.B2); }
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
    for (int o = threadIdx.x; o < L * L; o += 256) {
        const int i = o / L, j = o - i * L;
        float s = 0.f;
        for (int r = 0; r < nr; ++r) s = fmaf(fs[r * L + i], fs[r * L + j], s);
        part[(size_t)blockIdx.x * L * L + o] = s;
    }
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
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < c.n1; ++k) s1 += part[(size_t)k * LL + i];
        for (int k = 0; k < c.n2; ++k) s2 += part[(size_t)(c.n1 + k) * LL + i];
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

}  // namespace

extern "C" size_t nsvd_evd_scratch_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return 0;
    const Chunking c = chunking(B);
    return nsvd_align((size_t)(c.n1 + c.n2) * ((size_t)L * L + 1) * sizeof(float));
}

extern "C" int nsvd_evd_moments(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                                float* moments, void* scratch, void* stream) {
    if (!f || !Tf || !moments || !scratch || B <= 0 || L <= 0) return NSVD_EINVAL;
    if (L > MAXL) return NSVD_EUNSUPPORTED;
    if (mask_kind == NSVD_MASK_CUSTOM && !v) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    const Chunking c = chunking(B);
    const int nch = c.n1 + c.n2;
    float* part = (float*)scratch;
    float* part_op = part + (size_t)nch * L * L;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(evd_partial_kernel, dim3(nch), dim3(256), (size_t)CH * L * sizeof(float), s, f, Tf, B, L,
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
