// One Sketchy-style CDK training step as ONE C call (BASELINE configs[4]):
//   two towers forward -> normalize -> NestedLoRAForCDK loss -> its backward -> normalize backward -> two towers
//   backward -> clip_grad_norm_ over all 16 parameter tensors -> SGD with momentum, parameters updated in place.
// Replaces the loop body of examples/cdk/sketchy/main_sketchy.py:180-212 as scripts/exps/sketchy.sh configures it
// (--optimizer sgd --momentum 0.9 --clip_grad_norm), with the AMP branches off (this library computes in float32):
//   optimizer.zero_grad(); _, fx, _, fy = method(x, y); loss, *_ = method.compute_loss(fx, fy); loss.backward();
//   nn.utils.clip_grad_norm_(model.parameters(), max_norm); optimizer.step()
// (the scheduler's lr for the step is the caller's: it passes the already scheduled value, as for RMSprop).
// Composition of the library's own stages (tower.hip, row_normalize.hip, cdk_loss.hip) - no torch autograd, no
// torch.optim, no per-step allocation - plus the two kernels below: the squared gradient norm in a fixed summation
// order (per-block partials, then one block in double) and the clipped momentum update over every tensor in one pass.
#include <string.h>
#include "nsvd_kernels.h"

namespace {

constexpr int NT = 8;  // tensors per tower: W1 b1 g1 be1 W2 b2 g2 be2
constexpr int SUMSQ_BLOCKS = 64;    // blocks of the small-tensor pass

struct TensorTable {
    float* p[2 * NT];
    float* buf[2 * NT];
    unsigned short* h[2 * NT];  // null, or where the updated parameter is ALSO written as bfloat16 (mixed precision: the
                                // operand copies of W1 / W2 the next step's contractions read - no cast pass per step)
    int h_f16;                  // those copies are IEEE float16
    const float* g[2 * NT];
    size_t n[2 * NT];
    size_t start[2 * NT + 1];  // prefix sums in float4 groups (every tensor padded to a multiple of 4 in the tables)
};

// partial[b] = sum of squares of block b's share of the SMALL gradient tensors (biases, BatchNorm weights: the two
// weight matrices of a tower arrive as per-tile sums from the epilogues of their contractions, tower.hip)
__global__ void __launch_bounds__(256) cdk_sumsq_kernel(TensorTable t, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    for (int k = 0; k < 2 * NT; ++k) {
        if ((k % NT) == 0 || (k % NT) == 4) continue;  // W1, W2: summed by their GEMMs
        const float* g = t.g[k];
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < t.n[k]; i += (size_t)gridDim.x * 256)
            s = fmaf(g[i], g[i], s);
    }
    s = nsvd_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// g' = coef g; buf = first ? g' : momentum buf + g'; p -= lr buf   (torch.optim.SGD, no dampening / nesterov / decay)
// coef = torch's clip_grad_norm_ coefficient min(1, max_norm / (norm + 1e-6)), norm = sqrt of the sum of the partial
// sums of squares - added by EVERY workgroup for itself, in one fixed order (256 threads stride the partials, then a
// halving tree, in double): every workgroup gets the same bits, and the one-workgroup launch that used to sit between
// the gradient kernels and this one is gone. Workgroup 0 leaves norm and coefficient in scal / loss_out[3].
__global__ void __launch_bounds__(256) cdk_sgd_kernel(TensorTable t, const float* __restrict__ partial, int npartial,
                                                      float max_norm, float* __restrict__ scal,
                                                      float* __restrict__ loss_out, float lr, float momentum, int first,
                                                      NsvdCdkLossParts lp, const nsvd_grad_scaler* gs) {
    __shared__ double red[256];
    __shared__ float coef_s;
    // loss scaling (include/nsvd.h: nsvd_grad_scaler): the gradients carry the factor `scale`; the momentum buffers
    // start with the first step TAKEN. The state is only READ here (every workgroup the same values):
    // cdk_scaler_update_kernel advances it behind this kernel.
    float inv_scale = 1.f;
    if (gs) {
        inv_scale = 1.0f / gs->scale;
        first = gs->steps_ok == 0;
    }
    // the step's loss value: the loss kernels left per-block partials (no reduction launch of their own)
    if (blockIdx.x == 0 && loss_out && lp.part_op) nsvd_cdk_loss_sum(lp, reinterpret_cast<float*>(red), loss_out);
    __syncthreads();
    {
        double s = 0.0;
        for (int i = threadIdx.x; i < npartial; i += 256) s += (double)partial[i];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float norm_scaled = (float)sqrt(red[0]);
            const bool found_inf = gs && !(fabsf(norm_scaled) <= 3.0e38f);  // inf or NaN
            const float norm = norm_scaled * inv_scale;
            float coef = 1.f;
            if (max_norm > 0.f) {
                coef = max_norm / (norm + 1e-6f);
                coef = coef > 1.f ? 1.f : coef;  // (a NaN norm gives a NaN coefficient, as in torch)
            }
            coef_s = found_inf ? -1.f : coef;  // (a clip coefficient is never negative: -1 = skip the step)
            if (blockIdx.x == 0) {
                scal[0] = norm;
                scal[1] = coef;
                scal[2] = found_inf ? 1.f : 0.f;
                if (loss_out) loss_out[3] = norm;
            }
        }
        __syncthreads();
    }
    if (coef_s < 0.f) return;  // GradScaler.step(): inf / NaN gradients - optimizer.step() is skipped as a whole
    const float coef = coef_s;
    const size_t total4 = t.start[2 * NT];
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < total4; q += (size_t)gridDim.x * 256) {
        int k = 0;
        while (q >= t.start[k + 1]) ++k;
        const size_t e = (q - t.start[k]) * 4;
        const size_t left = t.n[k] - e;
        const float* g = t.g[k] + e;
        float* p = t.p[k] + e;
        float* b = t.buf[k] + e;
        if (left >= 4) {
            float4 gv = *reinterpret_cast<const float4*>(g);
            if (gs) {  // scaler.unscale_(): grad * (1 / scale), before the clip coefficient
                gv.x *= inv_scale; gv.y *= inv_scale; gv.z *= inv_scale; gv.w *= inv_scale;
            }
            float4 pv = *reinterpret_cast<float4*>(p);
            float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(b);
            bv.x = first ? gv.x * coef : fmaf(momentum, bv.x, gv.x * coef);
            bv.y = first ? gv.y * coef : fmaf(momentum, bv.y, gv.y * coef);
            bv.z = first ? gv.z * coef : fmaf(momentum, bv.z, gv.z * coef);
            bv.w = first ? gv.w * coef : fmaf(momentum, bv.w, gv.w * coef);
            pv.x = fmaf(-lr, bv.x, pv.x); pv.y = fmaf(-lr, bv.y, pv.y);
            pv.z = fmaf(-lr, bv.z, pv.z); pv.w = fmaf(-lr, bv.w, pv.w);
            *reinterpret_cast<float4*>(b) = bv;
            *reinterpret_cast<float4*>(p) = pv;
            if (t.h[k]) {
                typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                typedef float f2 __attribute__((ext_vector_type(2)));
                if (t.h_f16)
                    *reinterpret_cast<uint2*>(t.h[k] + e) =
                        make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((f2){pv.x, pv.y}, h2)),
                                   __builtin_bit_cast(unsigned, __builtin_convertvector((f2){pv.z, pv.w}, h2)));
                else
                    *reinterpret_cast<uint2*>(t.h[k] + e) =
                        make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((f2){pv.x, pv.y}, bf2)),
                                   __builtin_bit_cast(unsigned, __builtin_convertvector((f2){pv.z, pv.w}, bf2)));
            }
        } else {
            for (size_t c = 0; c < left; ++c) {
                const float gc = (gs ? g[c] * inv_scale : g[c]) * coef;
                const float bc = first ? gc : fmaf(momentum, b[c], gc);
                b[c] = bc;
                p[c] = fmaf(-lr, bc, p[c]);
            }
        }
    }
}

// GradScaler.update() (include/nsvd.h: nsvd_grad_scaler), one thread, behind the optimiser kernel that read the state
__global__ void cdk_scaler_update_kernel(nsvd_grad_scaler* gs, const float* __restrict__ scal) {
    const bool found_inf = scal[2] != 0.f;
    gs->last_found_inf = found_inf ? 1 : 0;
    if (found_inf) {
        gs->scale *= gs->backoff_factor;
        gs->growth_tracker = 0;
        gs->steps_skipped += 1;
    } else {
        gs->steps_ok += 1;
        if (++gs->growth_tracker == gs->growth_interval) {
            gs->scale *= gs->growth_factor;
            gs->growth_tracker = 0;
        }
    }
}

__global__ void cdk_scaler_init_kernel(nsvd_grad_scaler* gs, float scale, float growth, float backoff, int interval) {
    gs->scale = scale; gs->growth_factor = growth; gs->backoff_factor = backoff; gs->growth_interval = interval;
    gs->growth_tracker = 0; gs->steps_ok = 0; gs->steps_skipped = 0; gs->last_found_inf = 0;
}

struct StepWs {
    void* tower[2];
    void* cdk;
    float *z[2], *e[2], *ge[2], *dz[2];
    float* grad[2];  // per tower: [W1 | b1 | g1 | be1 | W2 | b2 | g2 | be2], each padded to 64 floats
    float* partial;  // [small tensors: SUMSQ_BLOCKS | tower x: per GEMM tile (+ per strip) | tower y: likewise]
    int npartial, nsmall;
    float* scal;
    float* narrow;   // scratch of the fused narrow end (cdk_narrow.hip), mixed precision
    size_t tower_bytes, cdk_bytes, bytes;
    size_t goff[NT], gn[NT];
};

StepWs carve_step(const nsvd_cdk_step_desc& d, void* base) {
    StepWs w;
    memset(&w, 0, sizeof(w));
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* q = p + off;
        off += nsvd_align(bytes);
        return q;
    };
    w.tower_bytes = nsvd_tower_workspace_bytes(d.B, d.d0, d.d1, d.d2);
    w.cdk_bytes = nsvd_cdk_workspace_bytes(d.B, d.d2, d.set_first_mode_const);
    for (int s = 0; s < 2; ++s) w.tower[s] = take(w.tower_bytes);
    w.cdk = take(w.cdk_bytes);
    const size_t bl = (size_t)d.B * d.d2 * sizeof(float);
    for (int s = 0; s < 2; ++s) {
        w.z[s] = (float*)take(bl);
        w.e[s] = (float*)take(bl);
        w.ge[s] = (float*)take(bl);
        w.dz[s] = (float*)take(bl);
    }
    const size_t n[NT] = {(size_t)d.d1 * d.d0, (size_t)d.d1, (size_t)d.d1, (size_t)d.d1,
                          (size_t)d.d2 * d.d1, (size_t)d.d2, (size_t)d.d2, (size_t)d.d2};
    size_t g = 0;
    for (int k = 0; k < NT; ++k) {
        w.goff[k] = g;
        w.gn[k] = n[k];
        g += (n[k] + 63) / 64 * 64;
    }
    for (int s = 0; s < 2; ++s) w.grad[s] = (float*)take(g * sizeof(float));
    // mixed precision with the fused narrow end: the small tensors' squares come from the kernels that write those
    // gradients (one float per strip of columns: d1 / 64 + nsvd_narrow_sumsq_count(d2) per tower)
    w.nsmall = (d.gemm_bf16 != 0 && nsvd_narrow_supported(2, d.B, d.d2)) ? (d.d1 / 64 + nsvd_narrow_sumsq_count(d.d2)) : 0;
    w.npartial = SUMSQ_BLOCKS + 2 * (nsvd_tower_sumsq_count(d.d0, d.d1, d.d2, d.gemm_bf16 != 0) + w.nsmall);
    w.partial = (float*)take((size_t)w.npartial * sizeof(float));
    w.scal = (float*)take(256);
    w.narrow = (float*)take(nsvd_narrow_scratch_floats(2, d.B, d.d2) * sizeof(float));
    w.bytes = off;
    return w;
}

bool desc_ok(const nsvd_cdk_step_desc* d) {
    if (!d) return false;
    if (d->grad_scaler && (d->gemm_bf16 == 0 || !nsvd_narrow_supported(2, d->B, d->d2))) return false;
    if ((d->gemm_bf16 & NSVD_TOWER16_F16) && !(d->gemm_bf16 & 1)) return false;
    if (nsvd_tower_workspace_bytes(d->B, d->d0, d->d1, d->d2) == 0) return false;
    if (d->gemm_bf16 != 0 && !nsvd_tower_mixed_supported(d->B, d->d0, d->d1, d->d2)) return false;
    if (d->normalize_mode != NSVD_NORMALIZE_L2_BALL && d->normalize_mode != NSVD_NORMALIZE_L2_SPHERE) return false;
    if (!(d->mu > 0.f)) return false;
    return true;
}

float* const* tower_fields(const nsvd_tower_params& t, float* out[NT]) {
    out[0] = t.W1; out[1] = t.b1; out[2] = t.g1; out[3] = t.be1;
    out[4] = t.W2; out[5] = t.b2; out[6] = t.g2; out[7] = t.be2;
    return out;
}

}  // namespace

extern "C" int nsvd_grad_scaler_init(void* state, float init_scale, float growth_factor, float backoff_factor,
                                     int growth_interval, void* stream) {
    if (!state || !(init_scale > 0.f) || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f) ||
        growth_interval < 1)
        return NSVD_EINVAL;
    hipLaunchKernelGGL(cdk_scaler_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (nsvd_grad_scaler*)state,
                       init_scale, growth_factor, backoff_factor, growth_interval);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t nsvd_cdk_step_workspace_bytes(const nsvd_cdk_step_desc* d) {
    if (!desc_ok(d)) return 0;
    return carve_step(*d, nullptr).bytes;
}

extern "C" int nsvd_cdk_step(const nsvd_cdk_step_desc* d, const float* x, const float* y,
                             const nsvd_tower_params* towers, const nsvd_tower_params* momentum_bufs, const float* v,
                             const float* M, float* loss, float* rs_joint, float* rs_indep, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!desc_ok(d) || !x || !y || !towers || !momentum_bufs || !v || !M || !loss || !ws) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    const StepWs w = carve_step(*d, ws);
    if (ws_bytes < w.bytes) return NSVD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, L = d->d2;
    const float r_up = sqrtf(d->mu);
    const float* in[2] = {x, y};
    int rc = 0;
    // forward: towers (BatchNorm running statistics updated, as a training-mode module does), normalisation
    const bool mixed = d->gemm_bf16 != 0;
    const nsvd_tower_params* tp[2] = {&towers[0], &towers[1]};
    // mixed precision: both towers through every launch together (tower.hip, mixed-precision section), and the narrow
    // end - split-K sum, second BatchNorm, normalisation - as three launches for both (cdk_narrow.hip)
    const bool narrow = mixed && nsvd_narrow_supported(2, B, L);
    NsvdTowerNarrowViews nv[2];
    if (mixed) {
        rc = nsvd_tower16_forward_pair(in, tp, B, d->d0, d->d1, d->d2, d->slope, d->bn_eps, d->bn_momentum, 1,
                                       d->gemm_bf16 | (narrow ? NSVD_TOWER16_WIDE_ONLY : 0), w.z, w.tower,
                                       w.tower_bytes, s);
        if (rc) return rc;
    }
    if (narrow) {
        NsvdNarrowFwd f;
        memset(&f, 0, sizeof(f));
        for (int t = 0; t < 2; ++t) {
            nv[t] = nsvd_tower16_narrow_views(2, B, d->d0, d->d1, d->d2, w.tower[t]);
            f.Y2p[t] = nv[t].Y2p; f.bias[t] = towers[t].b2; f.gamma[t] = towers[t].g2; f.beta[t] = towers[t].be2;
            f.running_mean[t] = towers[t].rm2; f.running_var[t] = towers[t].rv2;
            f.mean[t] = nv[t].mean2; f.invstd[t] = nv[t].inv2; f.Y2[t] = nv[t].Y2; f.z[t] = w.z[t]; f.e[t] = w.e[t];
        }
        f.nt = 2; f.B = B; f.N = L; f.S = nv[0].S; f.slice_stride = nv[0].slice_stride; f.part = w.narrow;
        f.eps = d->bn_eps; f.momentum = d->bn_momentum; f.r_up = r_up;
        f.sphere = d->normalize_mode == NSVD_NORMALIZE_L2_SPHERE;
        rc = nsvd_narrow_forward(f, s);
        if (rc) return rc;
    }
    for (int t = 0; t < 2 && !narrow; ++t) {
        if (!mixed) {
            rc = nsvd_tower_forward(in[t], &towers[t], B, d->d0, d->d1, d->d2, d->slope, d->bn_eps, d->bn_momentum, 1,
                                    0, w.z[t], w.tower[t], w.tower_bytes, stream);
            if (rc) return rc;
        }
        rc = nsvd_row_normalize_forward(w.z[t], B, L, r_up, d->normalize_mode, w.e[t], stream);
        if (rc) return rc;
    }
    // loss and its gradient w.r.t. the two embeddings
    NsvdCdkLossParts lparts;
    rc = nsvd_cdk_loss_forward_parts(w.e[0], w.e[1], nullptr, v, M, B, L, d->set_first_mode_const, rs_joint, rs_indep,
                                     w.cdk, w.cdk_bytes, &lparts, s);
    if (rc) return rc;
    rc = nsvd_cdk_loss_backward(v, B, L, d->set_first_mode_const, nullptr, w.ge[0], w.ge[1], w.cdk, w.cdk_bytes, stream);
    if (rc) return rc;
    // backward: normalisation, towers (gradients into the workspace)
    TensorTable tab;
    memset(&tab, 0, sizeof(tab));
    size_t q4 = 0;
    nsvd_tower_params gr[2];
    float* gp[2][NT];
    const int nsq = nsvd_tower_sumsq_count(d->d0, d->d1, d->d2, mixed);
    float* sq[2] = {w.partial + SUMSQ_BLOCKS, w.partial + SUMSQ_BLOCKS + nsq + w.nsmall};
    const bool small_fused = narrow && w.nsmall > 0;
    for (int t = 0; t < 2; ++t) {
        if (!narrow) {
            rc = nsvd_row_normalize_backward(w.z[t], w.ge[t], B, L, r_up, d->normalize_mode, w.dz[t], stream);
            if (rc) return rc;
        }
        nsvd_tower_params& g = gr[t];
        memset(&g, 0, sizeof(g));
        for (int k = 0; k < NT; ++k) gp[t][k] = w.grad[t] + w.goff[k];
        g.W1 = gp[t][0]; g.b1 = gp[t][1]; g.g1 = gp[t][2]; g.be1 = gp[t][3];
        g.W2 = gp[t][4]; g.b2 = gp[t][5]; g.g2 = gp[t][6]; g.be2 = gp[t][7];
        if (!mixed) {
            rc = nsvd_tower_backward_sumsq(in[t], &towers[t], w.dz[t], B, d->d0, d->d1, d->d2, d->slope, 0, &g,
                                           w.tower[t], w.tower_bytes, sq[t], stream);
            if (rc) return rc;
        }
    }
    if (narrow) {  // normalisation backward, second BatchNorm backward -> bfloat16 dY2 in the towers' workspaces
        NsvdNarrowBwd b;
        memset(&b, 0, sizeof(b));
        for (int t = 0; t < 2; ++t) {
            b.z[t] = w.z[t]; b.ge[t] = w.ge[t]; b.Y2[t] = nv[t].Y2; b.mean[t] = nv[t].mean2; b.invstd[t] = nv[t].inv2;
            b.gamma[t] = towers[t].g2; b.dz[t] = w.dz[t]; b.dY[t] = nv[t].dY2h;
            b.dgamma[t] = gr[t].g2; b.dbeta[t] = gr[t].be2; b.dbias[t] = gr[t].b2;
            b.sumsq[t] = small_fused ? sq[t] + nsq + d->d1 / 64 : nullptr;
        }
        b.nt = 2; b.B = B; b.N = L; b.dy_bf16 = (d->gemm_bf16 & NSVD_TOWER16_F16) ? 2 : 1; b.part = w.narrow; b.r_up = r_up;
        b.loss_scale = d->grad_scaler ? &((const nsvd_grad_scaler*)d->grad_scaler)->scale : nullptr;
        b.sphere = d->normalize_mode == NSVD_NORMALIZE_L2_SPHERE;
        rc = nsvd_narrow_backward(b, s);
        if (rc) return rc;
    }
    if (mixed) {
        const nsvd_tower_params* gq[2] = {&gr[0], &gr[1]};
        const float* dzs[2] = {w.dz[0], w.dz[1]};
        rc = nsvd_tower16_backward_pair(in, tp, dzs, B, d->d0, d->d1, d->d2, d->slope, gq, w.tower, w.tower_bytes, sq, s,
                                        (narrow ? NSVD_TOWER16_WIDE_ONLY : 0) | (small_fused ? NSVD_TOWER16_SMALL_SUMSQ : 0) |
                                            (d->gemm_bf16 & NSVD_TOWER16_F16));
        if (rc) return rc;
    }
    for (int t = 0; t < 2; ++t) {
        float *pp[NT], *bb[NT];
        tower_fields(towers[t], pp);
        tower_fields(momentum_bufs[t], bb);
        void *w1h = nullptr, *w2h = nullptr;
        if (mixed) nsvd_tower16_weight_copies(B, d->d0, d->d1, d->d2, w.tower[t], &w1h, &w2h);
        for (int k = 0; k < NT; ++k) {
            const int i = t * NT + k;
            if (!pp[k] || !bb[k]) return NSVD_EINVAL;
            if ((((uintptr_t)pp[k] | (uintptr_t)bb[k]) & 15) != 0) return NSVD_EINVAL;
            tab.p[i] = pp[k]; tab.buf[i] = bb[k]; tab.g[i] = gp[t][k]; tab.n[i] = w.gn[k];
            tab.h[i] = k == 0 ? (unsigned short*)w1h : (k == 4 ? (unsigned short*)w2h : nullptr);
            tab.start[i] = q4;
            q4 += (w.gn[k] + 3) / 4;
        }
    }
    tab.start[2 * NT] = q4;
    tab.h_f16 = (d->gemm_bf16 & NSVD_TOWER16_F16) ? 1 : 0;
    // clip_grad_norm_ + SGD momentum over all 16 tensors
    if (!small_fused) {
        hipLaunchKernelGGL(cdk_sumsq_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, tab, w.partial);
        NSVD_CHECK_LAUNCH();
    }
    // (small_fused: the SUMSQ_BLOCKS slots are unused - the partials the optimiser adds start behind them)
    const float* part0 = small_fused ? w.partial + SUMSQ_BLOCKS : w.partial;
    const int npart = small_fused ? w.npartial - SUMSQ_BLOCKS : w.npartial;
    size_t blocks = (q4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;  // (each workgroup first adds the partials for itself: not too many of them)
    hipLaunchKernelGGL(cdk_sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, tab, part0, npart,
                       (float)d->max_grad_norm, w.scal, loss, (float)d->lr, (float)d->momentum, d->first_step, lparts,
                       (const nsvd_grad_scaler*)d->grad_scaler);
    NSVD_CHECK_LAUNCH();
    if (d->grad_scaler) {
        hipLaunchKernelGGL(cdk_scaler_update_kernel, dim3(1), dim3(1), 0, s, (nsvd_grad_scaler*)d->grad_scaler, w.scal);
        NSVD_CHECK_LAUNCH();
    }
    return 0;
}
