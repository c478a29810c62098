// Importance-weighted central-difference Hamiltonian on the head outputs, and the head of its
// backward.  Generic-path version (the fused MFMA kernel has the same math in its epilogue:
// nsvd_fd_point below is shared).
//   reference: WaveFunctions.forward           examples/operator/pde/__init__.py:15-16
//              ExponentialMask.forward         examples/operator/pde/boundary.py:46-53
//              VectorizedLaplacian.__call__    examples/operator/pde/diff_ops.py:9-23, 25-52
//              NegativeHamiltonian.__call__    examples/operator/pde/schrodinger/__init__.py:16-22
//              OperatorWrapper.__call__        examples/__init__.py:7-9
#include "nsvd_kernels.h"
#include "fd_math.h"

namespace {

__global__ void __launch_bounds__(256) fd_epilogue_kernel(const float* __restrict__ base, int ldr,
                                                          const float* __restrict__ x,
                                                          const float* __restrict__ scales, nsvd_problem prob,
                                                          float log_norm, int B, int D, int L, float* __restrict__ f,
                                                          float* __restrict__ Tf, float* __restrict__ jac,
                                                          float* __restrict__ dsc, int evenodd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * L) return;
    const int b = idx / L, l = idx - b * L;
    float xc[NSVD_FD_MAXD];
    for (int d = 0; d < D; ++d) xc[d] = x[(size_t)b * D + d];
    const int E = 1 + 2 * D;
    float bv[2 * NSVD_FD_MAXD + 1];
    for (int e = 0; e < E; ++e) bv[e] = base[(size_t)l * ldr + (size_t)e * B + b];
    const float s_l = scales ? scales[l] : 0.f;
    NsvdFdOut o;
    if (evenodd) {  // rows 1 + 2 d / 2 + 2 d hold the even / odd perturbations of direction d (fused split-stencil form)
        float bE[NSVD_FD_MAXD], bO[NSVD_FD_MAXD];
        for (int d = 0; d < D; ++d) {
            bE[d] = bv[1 + 2 * d];
            bO[d] = bv[2 + 2 * d];
        }
        o = nsvd_fd_evenodd(bv[0], bE, bO, xc, D, scales != nullptr, s_l, prob, log_norm);
    } else {
        o = nsvd_fd_point(bv, xc, D, scales != nullptr, s_l, prob, log_norm);
    }
    f[idx] = o.f;
    Tf[idx] = o.Tf;
    if (jac) jac[idx] = o.jac;
    if (dsc) dsc[idx] = o.dsc;
}

__global__ void __launch_bounds__(256) head_backward_kernel(const float* __restrict__ df,
                                                            const float* __restrict__ jac, int B, int L,
                                                            float* __restrict__ dzT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * L) return;
    const int l = idx / B, b = idx - l * B;
    dzT[idx] = df[(size_t)b * L + l] * jac[(size_t)b * L + l];
}

__global__ void __launch_bounds__(256) dscales_kernel(const float* __restrict__ df, const float* __restrict__ dsc,
                                                      int B, int L, float* __restrict__ dscales) {
    __shared__ float red[4];
    const int l = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) s += df[(size_t)b * L + l] * dsc[(size_t)b * L + l];
    s = nsvd_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dscales[l] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) model_out_kernel(const float* __restrict__ base, int ldr,
                                                        const float* __restrict__ x,
                                                        const float* __restrict__ scales, float c, int B, int D, int L,
                                                        float* __restrict__ out, float* __restrict__ jac,
                                                        float* __restrict__ dsc) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * L) return;
    const int b = idx / L, l = idx - b * L;
    const float bv = base[(size_t)l * ldr + b];
    float mk = 1.f, r = 0.f;
    if (scales) {
        float r2 = 0.f;
        for (int d = 0; d < D; ++d) r2 = fmaf(x[(size_t)b * D + d], x[(size_t)b * D + d], r2);
        r = sqrtf(r2);
        mk = expf(-r / scales[l]);
    }
    out[idx] = c * bv * mk;
    if (jac) jac[idx] = c * mk;                                              // d out / d base
    if (dsc) dsc[idx] = scales ? c * bv * mk * r / (scales[l] * scales[l]) : 0.f;  // d out / d scales_l
}

}  // namespace

int nsvd_model_out(const float* base, int ldr, const float* x, const float* scales, float c, int B, int D, int L,
                   float* out, float* jac, float* dsc, hipStream_t s) {
    hipLaunchKernelGGL(model_out_kernel, dim3(nsvd_cdiv(B * L, 256)), dim3(256), 0, s, base, ldr, x, scales, c, B, D,
                       L, out, jac, dsc);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_fd_epilogue(const float* base, int ldr, const float* x, const float* scales, const nsvd_problem& prob,
                     int B, int D, int L, float* f, float* Tf, float* jac, float* dsc, hipStream_t s, int evenodd) {
    if (D > NSVD_FD_MAXD) return NSVD_EUNSUPPORTED;
    const float log_norm = nsvd_gauss_log_norm(D, prob.sigma);
    hipLaunchKernelGGL(fd_epilogue_kernel, dim3(nsvd_cdiv(B * L, 256)), dim3(256), 0, s, base, ldr, x, scales, prob,
                       log_norm, B, D, L, f, Tf, jac, dsc, evenodd);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_head_backward(const float* df, const float* jac, const float* dsc, int B, int L, float* dzT,
                       float* dscales, hipStream_t s) {
    hipLaunchKernelGGL(head_backward_kernel, dim3(nsvd_cdiv(B * L, 256)), dim3(256), 0, s, df, jac, B, L, dzT);
    NSVD_CHECK_LAUNCH();
    if (dscales) {
        hipLaunchKernelGGL(dscales_kernel, dim3(L), dim3(256), 0, s, df, dsc, B, L, dscales);
        NSVD_CHECK_LAUNCH();
    }
    return 0;
}
