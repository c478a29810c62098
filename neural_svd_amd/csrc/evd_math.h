// NestedLoRA EVD loss pieces shared by evd_loss.hip (stand-alone loss kernels) and pmlp_bwd.hip (the
// backward chain kernel evaluates d loss / d f per sample itself when handed the moments).
//   reference: methods/nestedlora.py:40-54 (masks), :57-64 (metric), :70-111 (loss forward / backward)
#pragma once
#include "nsvd_common.h"

constexpr int NSVD_EVD_CH = 64;  // rows per partial-moment chunk

struct NsvdEvdChunking {
    int B1, B2, n1, n2;
};
__host__ __device__ inline NsvdEvdChunking nsvd_evd_chunking(int B) {
    NsvdEvdChunking c;
    c.B1 = (B + 1) / 2;  // torch.chunk: first half gets the ceil
    c.B2 = B - c.B1;
    c.n1 = (c.B1 + NSVD_EVD_CH - 1) / NSVD_EVD_CH;
    c.n2 = (c.B2 + NSVD_EVD_CH - 1) / NSVD_EVD_CH;
    return c;
}

__device__ __forceinline__ float nsvd_mask_v(int kind, const float* v, int l, int L) {
    if (kind == NSVD_MASK_SEQUENTIAL) return 1.f;
    if (kind == NSVD_MASK_JOINT) return (float)(L - l) / (float)L;
    return v[l];
}
__device__ __forceinline__ float nsvd_mask_M(int kind, const float* M, int l, int m, int L) {
    if (kind == NSVD_MASK_SEQUENTIAL) return l <= m ? 1.f : 0.f;
    if (kind == NSVD_MASK_JOINT) return (float)(L - max(l, m)) / (float)L;  // min(v_l, v_m)
    return M[l * L + m];
}

// Everything a kernel needs to evaluate the loss and its gradient w.r.t. f for its own rows.
struct NsvdEvdIn {
    const float* f;         // (B, L)
    const float* Tf;        // (B, L)
    const float* v;         // custom masks, or null for the generated kinds
    const float* M;
    const float* moments;   // reduced (2 L^2 + 1), e.g. after the data-parallel all-reduce; or null:
    const float* part;      // per-chunk partial sums [n1 + n2][L*L] of evd_partial_kernel
    const float* part_op;   // [n1 + n2]
    float* moments_out;     // block 0 stores the reduced moments here (may be null)
    float* loss;            // block 0 stores {loss, operator term, metric term} (may be null)
    float grad_scale;
    int kind;
    int Lg;     // number of heads of f / Tf / the moments (>= the caller's local head count)
    int l_off;  // first global head of the caller (head-parallel sharding); 0 otherwise
};

__device__ __forceinline__ float nsvd_evd_mask_v(const NsvdEvdIn& in, int l, int L) {
    return nsvd_mask_v(in.kind, in.v, l, L);
}
__device__ __forceinline__ float nsvd_evd_mask_M(const NsvdEvdIn& in, int l, int m, int L) {
    return nsvd_mask_M(in.kind, in.M, l, m, L);
}

// lam_f{1,2}[idx] (h = 0 / 1): from the reduced vector, or the chunk partials summed in chunk order
__device__ __forceinline__ float nsvd_evd_lam(const NsvdEvdIn& in, int h, int idx, int B, int L) {
    const int LL = L * L;
    if (in.moments) return in.moments[h * LL + idx];
    const NsvdEvdChunking c = nsvd_evd_chunking(B);
    const int k0 = h ? c.n1 : 0, k1 = h ? c.n1 + c.n2 : c.n1;
    float s = 0.f;
    for (int k = k0; k < k1; ++k) s += in.part[(size_t)k * LL + idx];
    return s / (float)(h ? c.B2 : c.B1);
}

// Called by ALL 256 threads of ONE block: loss scalars (+ the reduced moments). red: >= 4 floats of LDS.
__device__ __forceinline__ void nsvd_evd_finish(const NsvdEvdIn& in, int B, int L, float* red) {
    const int LL = L * L, tid = threadIdx.x;
    float s = 0.f;
    for (int o = tid; o < LL; o += 256) {
        const float l1 = nsvd_evd_lam(in, 0, o, B, L), l2 = nsvd_evd_lam(in, 1, o, B, L);
        s = fmaf(nsvd_evd_mask_M(in, o / L, o % L, L) * l1, l2, s);
        if (in.moments_out) {
            in.moments_out[o] = l1;
            in.moments_out[LL + o] = l2;
        }
    }
    s = nsvd_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float opm;
        if (in.moments) {
            opm = in.moments[2 * LL];
        } else {
            const NsvdEvdChunking c = nsvd_evd_chunking(B);
            opm = 0.f;
            for (int k = 0; k < c.n1 + c.n2; ++k) opm += in.part_op[k];
            opm /= (float)B;
        }
        const float metric = (red[0] + red[1]) + (red[2] + red[3]);
        if (in.loss) {
            in.loss[0] = -2.f * opm + metric;
            in.loss[1] = -2.f * opm;
            in.loss[2] = metric;
        }
        if (in.moments_out) in.moments_out[2 * LL] = opm;
    }
}
