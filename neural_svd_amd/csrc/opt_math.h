// RMSprop + EMA update of one parameter, shared by the stand-alone optimiser kernel (optimizer.hip) and the
// weight-gradient kernel's fused epilogue (pmlp_bwd.hip).
//   reference: torch.optim.RMSprop as configured at examples/utils.py:50-57 (alpha, eps = 1e-10, momentum 0,
//              not centred), stepped at examples/operator/__init__.py:69-70; torch_ema update at :73.
#pragma once
#include "nsvd_common.h"

struct NsvdHyper {
    float lr, alpha, one_minus_alpha, eps, one_minus_decay, grad_scale;
};

// parameter / RMSprop square average / EMA shadow (null: none) of one tensor, element-aligned with its gradient
struct NsvdOptPtrs {
    float* p;
    float* sq;
    float* ema;
};

// optimiser step fused into the backward: state tensors in the parameters' layouts
struct NsvdOptStep {
    NsvdHyper h;
    nsvd_params sq;
    const nsvd_params* ema;  // null: no EMA
    nsvd_step_state* state;  // device-resident schedule (nsvd.h): h is then read from state->cur on the device
    int emit_planes;         // NSVD_PATH_FUSED_BF16X3 steps: the weight-gradient epilogue also writes the bf16 planes of
                             // the UPDATED hidden-layer weights (pmlp_layer0_bf3.h) into the workspace the next forward reads
};

static_assert(sizeof(((nsvd_step_state*)0)->cur) == sizeof(NsvdHyper), "nsvd_step_state::cur is an NsvdHyper");
__device__ __forceinline__ const NsvdHyper* nsvd_state_hyper(const nsvd_step_state* st) {
    return reinterpret_cast<const NsvdHyper*>(&st->cur);
}

// cur <- the scheduled values of step st->step: CosineAnnealingLR's closed form after `step` scheduler steps
// (torch/optim/lr_scheduler.py; examples/operator/__init__.py:35,71-72) and torch_ema's decay at its (step + 1)-th
// update (:36,73), in the double-precision expressions trainer.cosine_lr / FusedTrainer._advance_schedule evaluate on
// the host, rounded to float32 where nsvd_make_hyper rounds. Called by ONE thread.
__device__ inline void nsvd_step_state_derive(nsvd_step_state* st) {
#pragma clang fp contract(off)
    const unsigned long long t = st->step;
    double lr = st->lr0;
    if (st->T_max) {
        const double c = cos((3.141592653589793 * (double)t) / (double)st->T_max);
        lr = st->eta_min + ((st->lr0 - st->eta_min) * (1.0 + c)) / 2.0;
    }
    const double n = (double)(t + 1);
    const double warm = (1.0 + n) / (10.0 + n);
    const double decay = st->ema_decay < warm ? st->ema_decay : warm;
    st->cur.lr = (float)lr;
    st->cur.alpha = (float)st->alpha;
    st->cur.one_minus_alpha = (float)(1.0 - st->alpha);
    st->cur.eps = (float)st->eps;
    st->cur.one_minus_decay = (float)(1.0 - decay);
    st->cur.grad_scale = 1.0f;
}

// host scalars are doubles (Python floats), rounded to float32 exactly where torch rounds them
static inline NsvdHyper nsvd_make_hyper(double lr, double alpha, double eps, double ema_decay, double grad_scale) {
    NsvdHyper h;
    h.lr = (float)lr;
    h.alpha = (float)alpha;
    h.one_minus_alpha = (float)(1.0 - alpha);
    h.eps = (float)eps;
    h.one_minus_decay = (float)(1.0 - ema_decay);
    h.grad_scale = (float)grad_scale;
    return h;
}

// ema by reference + flag (not an optional pointer: a conditionally taken address of a local keeps it in scratch)
__device__ __forceinline__ void nsvd_rmsprop_upd(float& p, float g, float& sq, float& ema, bool has_ema,
                                                 const NsvdHyper& h) {
    const float lr = h.lr, eps = h.eps, one_minus_decay = h.one_minus_decay;
    g *= h.grad_scale;
    sq = h.alpha * sq + h.one_minus_alpha * (g * g);  // square_avg.mul_(alpha).addcmul_(g, g, value=1-alpha)
    const float avg = sqrtf(sq) + eps;                // square_avg.sqrt().add_(eps)
    p = p - lr * (g / avg);                           // param.addcdiv_(g, avg, value=-lr)
    if (has_ema) ema = ema - one_minus_decay * (ema - p);
}
