// Fused RMSprop + EMA over a flat parameter buffer: one pass, 16-byte accesses.
//   reference: torch.optim.RMSprop as configured at examples/utils.py:50-57 (alpha, eps=1e-10,
//              momentum 0, not centred), stepped at examples/operator/__init__.py:69-70;
//              torch_ema update at examples/operator/__init__.py:73.
// HBM-bound: reads p, g, sq, ema and writes p, sq, ema = 28 B per parameter.
#include <string.h>
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
                                                          size_t n, Hyper h, nsvd_step_state* state, int advance) {
    if (state) {  // device-resident schedule: the step's values from state->cur (grad_scale stays the caller's)
        const float gs = h.grad_scale;
        h = *nsvd_state_hyper(state);
        h.grad_scale = gs;
    }
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
    // last optimiser launch of the step: nothing in this kernel reads `step` (only `cur`, which the NEXT step's first
    // kernel rewrites after the kernel boundary)
    if (state && advance && blockIdx.x == 0 && threadIdx.x == 0) state->step += 1;
}

// The same update over a TABLE of tensors in one launch (the generic path's step: one launch instead of one per tensor -
// nine launches of ~5 us each at three hidden layers). Element for element the arithmetic of the kernel above.
template <bool HAS_EMA>
__global__ void __launch_bounds__(256) rmsprop_ema_table_kernel(NsvdOptTable t, Hyper h) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t total4 = t.start[t.count];
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += stride) {
        int k = 0;
        while (q >= t.start[k + 1]) ++k;
        const size_t e = (q - t.start[k]) * 4;
        const size_t left = t.n[k] - e;
        float* p = t.p[k] + e;
        const float* g = t.g[k] + e;
        float* sq = t.sq[k] + e;
        float* ema = HAS_EMA ? t.ema[k] + e : nullptr;
        if (left >= 4) {
            float4 pv = *reinterpret_cast<float4*>(p);
            const float4 gv = *reinterpret_cast<const float4*>(g);
            float4 sv = *reinterpret_cast<float4*>(sq);
            float4 ev = HAS_EMA ? *reinterpret_cast<float4*>(ema) : make_float4(0, 0, 0, 0);
            upd(pv.x, gv.x, sv.x, HAS_EMA ? &ev.x : nullptr, h);
            upd(pv.y, gv.y, sv.y, HAS_EMA ? &ev.y : nullptr, h);
            upd(pv.z, gv.z, sv.z, HAS_EMA ? &ev.z : nullptr, h);
            upd(pv.w, gv.w, sv.w, HAS_EMA ? &ev.w : nullptr, h);
            *reinterpret_cast<float4*>(p) = pv;
            *reinterpret_cast<float4*>(sq) = sv;
            if (HAS_EMA) *reinterpret_cast<float4*>(ema) = ev;
        } else {
            for (size_t c = 0; c < left; ++c) {
                float pv = p[c], sv = sq[c], ev = HAS_EMA ? ema[c] : 0.f;
                upd(pv, g[c], sv, HAS_EMA ? &ev : nullptr, h);
                p[c] = pv;
                sq[c] = sv;
                if (HAS_EMA) ema[c] = ev;
            }
        }
    }
}

__global__ void step_state_init_kernel(nsvd_step_state* st, nsvd_step_state v) {
    *st = v;
    nsvd_step_state_derive(st);
}
__global__ void step_state_begin_kernel(nsvd_step_state* st) { nsvd_step_state_derive(st); }

}  // namespace

int nsvd_rmsprop_launch(float* p, const float* grad, float* sq, float* ema, size_t n, const NsvdHyper& h,
                        hipStream_t s, nsvd_step_state* state, int advance) {
    if (n == 0) return 0;
    const uintptr_t al = (uintptr_t)p | (uintptr_t)grad | (uintptr_t)sq | (uintptr_t)ema;
    const size_t n4 = (al & 15) ? 0 : n / 4;  // unaligned (never with torch allocations): scalar path
    const size_t work = n4 ? n4 : n;
    size_t blocks = (work + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (ema) hipLaunchKernelGGL(rmsprop_ema_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, p, grad, sq, ema,
                                n4, n, h, state, advance);
    else hipLaunchKernelGGL(rmsprop_ema_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, p, grad, sq,
                            (float*)nullptr, n4, n, h, state, advance);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_rmsprop_table_launch(NsvdOptTable& t, const NsvdHyper& h, hipStream_t s) {
    if (t.count <= 0 || t.count > NSVD_OPT_TABLE_MAX) return NSVD_EINVAL;
    bool has_ema = t.ema[0] != nullptr;
    size_t q4 = 0;
    for (int k = 0; k < t.count; ++k) {
        if (!t.p[k] || !t.g[k] || !t.sq[k] || (t.ema[k] != nullptr) != has_ema) return NSVD_EINVAL;
        const uintptr_t al = (uintptr_t)t.p[k] | (uintptr_t)t.g[k] | (uintptr_t)t.sq[k] | (uintptr_t)t.ema[k];
        if (al & 15) return NSVD_EUNSUPPORTED;  // (never with torch allocations; the caller launches per tensor then)
        t.start[k] = q4;
        q4 += (t.n[k] + 3) / 4;
    }
    t.start[t.count] = q4;
    if (q4 == 0) return 0;
    size_t blocks = (q4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (has_ema) hipLaunchKernelGGL(rmsprop_ema_table_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, t, h);
    else hipLaunchKernelGGL(rmsprop_ema_table_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, t, h);
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

extern "C" int nsvd_rmsprop_ema_step_dev(float* p, const float* grad, float* sq, float* ema, size_t n,
                                         nsvd_step_state* state, double grad_scale, int advance, void* stream) {
    if (!p || !grad || !sq || !state || ((uintptr_t)state & 7) != 0) return NSVD_EINVAL;
    return nsvd_rmsprop_launch(p, grad, sq, ema, n, nsvd_make_hyper(0.0, 0.0, 0.0, 0.0, grad_scale),
                               (hipStream_t)stream, state, advance);
}

extern "C" int nsvd_step_state_init(nsvd_step_state* state, double lr0, double eta_min, unsigned long long T_max,
                                    double alpha, double eps, double ema_decay, unsigned long long step,
                                    void* stream) {
    if (!state || ((uintptr_t)state & 7) != 0) return NSVD_EINVAL;
    nsvd_step_state v;
    memset(&v, 0, sizeof(v));
    v.step = step;
    v.T_max = T_max;
    v.lr0 = lr0; v.eta_min = eta_min; v.alpha = alpha; v.eps = eps; v.ema_decay = ema_decay;
    hipLaunchKernelGGL(step_state_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, v);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_step_state_begin(nsvd_step_state* state, void* stream) {
    if (!state || ((uintptr_t)state & 7) != 0) return NSVD_EINVAL;
    hipLaunchKernelGGL(step_state_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    NSVD_CHECK_LAUNCH();
    return 0;
}
