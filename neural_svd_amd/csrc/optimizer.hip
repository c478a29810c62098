// Fused RMSprop + EMA over a flat parameter buffer: one pass, 16-byte accesses.
//   reference: torch.optim.RMSprop as configured at examples/utils.py:50-57 (alpha, eps=1e-10,
//              momentum 0, not centred), stepped at examples/operator/__init__.py:69-70;
//              torch_ema update at examples/operator/__init__.py:73.
// HBM-bound: reads p, g, sq, ema and writes p, sq, ema = 28 B per parameter.
#include "nsvd_kernels.h"
#include "opt_math.h"

namespace {

typedef NsvdHyper Hyper;
__device__ __forceinline__ void upd(float& p, float g, float& sq, float* ema, const Hyper& h) {
    float none = 0.f;
    nsvd_rmsprop_upd(p, g, sq, ema ? *ema : none, ema != nullptr, h);
}

template <bool HAS_EMA>
__global__ void __launch_bounds__(256) rmsprop_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ sq, float* __restrict__ ema, size_t n4,
                                                          size_t n, Hyper h) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 sv = reinterpret_cast<float4*>(sq)[i];
        float4 ev = HAS_EMA ? reinterpret_cast<float4*>(ema)[i] : make_float4(0, 0, 0, 0);
        upd(pv.x, gv.x, sv.x, HAS_EMA ? &ev.x : nullptr, h);
        upd(pv.y, gv.y, sv.y, HAS_EMA ? &ev.y : nullptr, h);
        upd(pv.z, gv.z, sv.z, HAS_EMA ? &ev.z : nullptr, h);
        upd(pv.w, gv.w, sv.w, HAS_EMA ? &ev.w : nullptr, h);
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(sq)[i] = sv;
        if (HAS_EMA) reinterpret_cast<float4*>(ema)[i] = ev;
    }
    // tail (n not a multiple of 4, or unaligned buffers: n4 == 0 and everything goes through here)
    for (size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
        float pv = p[t], sv = sq[t], ev = HAS_EMA ? ema[t] : 0.f;
        upd(pv, g[t], sv, HAS_EMA ? &ev : nullptr, h);
        p[t] = pv;
        sq[t] = sv;
        if (HAS_EMA) ema[t] = ev;
    }
}

}  // namespace

int nsvd_rmsprop_launch(float* p, const float* grad, float* sq, float* ema, size_t n, const NsvdHyper& h,
                        hipStream_t s) {
    if (n == 0) return 0;
    const uintptr_t al = (uintptr_t)p | (uintptr_t)grad | (uintptr_t)sq | (uintptr_t)ema;
    const size_t n4 = (al & 15) ? 0 : n / 4;  // unaligned (never with torch allocations): scalar path
    const size_t work = n4 ? n4 : n;
    size_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (ema) hipLaunchKernelGGL(rmsprop_ema_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p, grad, sq, ema,
                                n4, n, h);
    else hipLaunchKernelGGL(rmsprop_ema_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p, grad, sq,
                            (float*)nullptr, n4, n, h);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_rmsprop_ema_step(float* p, const float* grad, float* sq, float* ema, size_t n, double lr,
                                     double alpha, double eps, double ema_decay, double grad_scale,
                                     void* stream) {
    if (!p || !grad || !sq) return NSVD_EINVAL;
    return nsvd_rmsprop_launch(p, grad, sq, ema, n, nsvd_make_hyper(lr, alpha, eps, ema_decay, grad_scale),
                               (hipStream_t)stream);
}
