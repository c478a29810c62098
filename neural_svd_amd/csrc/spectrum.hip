// Streaming accumulators of compute_spectrum_evd (reference methods/spectrum.py:56-75):
//   cov += phi^T phi, quad += phi^T Tphi  with phi = nan_to_num(w f), Tphi = nan_to_num(w Tf),
//   w = sqrt(p_train(x)) / sqrt(p_val), rows with x ~ 0 zeroed in Tphi (:73).
// One workgroup per 128 rows: rows staged in LDS, L x L partial products in registers, then one
// atomic per (i, j) and workgroup. Acc = float: the reference's float32 accumulators; Acc = double
// (nsvd_spectrum_accumulate_f64): products and sums in float64 - the quotient diag(quad) / diag(cov) of an excited state
// is a sum of terms up to 10^2 x larger than itself (the potential near the nucleus against the kinetic term), and a
// float32 running sum over 500 workgroups then carries it to 1e-4 only (measured at configs[1]: 1.7e-4 on the n = 4
// shell with float32 accumulators, 1e-6 with these - scripts/dev/parity_diag.py).
#include "nsvd_kernels.h"
#include <float.h>

namespace {

constexpr int SR = 128;     // rows per workgroup
constexpr int SMAXL = 64;

__device__ __forceinline__ float nan_to_num(float v) {
    if (isnan(v)) return 0.f;
    if (isinf(v)) return v > 0 ? FLT_MAX : -FLT_MAX;
    return v;
}

template <typename Acc>
__global__ void __launch_bounds__(256) spectrum_kernel(const float* __restrict__ f, const float* __restrict__ Tf,
                                                       const float* __restrict__ x, int B, int L, int D, float sigma,
                                                       float log_norm, int use_imp, float inv_sqrt_val, int pad,
                                                       Acc* __restrict__ cov, Acc* __restrict__ quad) {
    // pad = 1 (set_first_mode_const, methods/spectrum.py:68-70): a constant-one column in FRONT of the weighted phi and
    // Tphi (ConstantPad1d((1, 0), 1) after the weighting); the accumulators are then (L + 1, L + 1)
    extern __shared__ __attribute__((aligned(16))) float sm[];  // phi[SR][L + pad], tphi[SR][L + pad]
    const int Lin = L;
    L += pad;
    float* ph = sm;
    float* tp = sm + SR * L;
    const int r0 = blockIdx.x * SR;
    const int nr = min(SR, B - r0);
    for (int i = threadIdx.x; i < nr * L; i += 256) {
        const int r = i / L, c = i - r * L;
        const float* xr = x + (size_t)(r0 + r) * D;
        const float sp = use_imp ? nsvd_sqrt_gauss_pdf(xr, D, sigma, log_norm) : 1.f;
        const float w = sp * inv_sqrt_val;
        bool zero = true;
        for (int d = 0; d < D; ++d) zero = zero && (fabsf(xr[d]) <= 1e-8f);  // torch.isclose(x, 0)
        const bool one = pad && c == 0;
        const size_t src = (size_t)(r0 + r) * Lin + (c - pad);
        ph[i] = one ? 1.f : nan_to_num(w * f[src]);
        const float t = one ? 1.f : nan_to_num(w * Tf[src]);
        tp[i] = zero ? 0.f : t;
    }
    __syncthreads();
    for (int o = threadIdx.x; o < L * L; o += 256) {
        const int i = o / L, j = o - i * L;
        Acc c = 0, q = 0;
        for (int r = 0; r < nr; ++r) {
            const Acc pi = ph[r * L + i];
            c = fma(pi, (Acc)ph[r * L + j], c);
            q = fma(pi, (Acc)tp[r * L + j], q);
        }
        atomicAdd(&cov[o], c);
        atomicAdd(&quad[o], q);
    }
}

}  // namespace

template <typename Acc>
static int spectrum_accumulate_impl(const float* f, const float* Tf, const float* x, int B, int L, int D, float sigma,
                                    int use_importance, float lim, Acc* cov, Acc* quad, void* stream, int pad = 0) {
    if (!f || !Tf || !x || !cov || !quad || B <= 0 || L <= 0 || D <= 0 || pad < 0 || pad > 1) return NSVD_EINVAL;
    if (L > SMAXL) return NSVD_EUNSUPPORTED;
    // importance_val is built as a float32 tensor in the reference (main_pde.py:130)
    const float pval = (float)(1.0 / pow(2.0 * (double)lim, (double)D));
    const float inv_sqrt_val = 1.f / sqrtf(pval);
    const size_t lds = (size_t)2 * SR * (L + pad) * sizeof(float);
    hipLaunchKernelGGL(spectrum_kernel<Acc>, dim3(nsvd_cdiv(B, SR)), dim3(256), lds, (hipStream_t)stream, f, Tf, x, B,
                       L, D, sigma, nsvd_gauss_log_norm(D, sigma), use_importance, inv_sqrt_val, pad, cov, quad);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_spectrum_accumulate(const float* f, const float* Tf, const float* x, int B, int L, int D,
                                        float sigma, int use_importance, float lim, float* cov, float* quad,
                                        void* stream) {
    return spectrum_accumulate_impl<float>(f, Tf, x, B, L, D, sigma, use_importance, lim, cov, quad, stream);
}

extern "C" int nsvd_spectrum_accumulate_f64(const float* f, const float* Tf, const float* x, int B, int L, int D,
                                            float sigma, int use_importance, float lim, double* cov, double* quad,
                                            void* stream) {
    return spectrum_accumulate_impl<double>(f, Tf, x, B, L, D, sigma, use_importance, lim, cov, quad, stream);
}

extern "C" int nsvd_spectrum_accumulate_const_f64(const float* f, const float* Tf, const float* x, int B, int L, int D,
                                                  float sigma, int use_importance, float lim, double* cov,
                                                  double* quad, void* stream) {
    return spectrum_accumulate_impl<double>(f, Tf, x, B, L, D, sigma, use_importance, lim, cov, quad, stream, 1);
}
