/*
 * nsvd.h - C ABI of libnsvd_hip.so: the MI355X (gfx950) NestedLoRA / NeuralSVD PDE training step.
 *
 * This is the drop-in boundary for ONE path of jongharyu/neural-svd (all citations are paths in
 * that repository):
 *
 *     loss, aux = method.compute_loss_operator(operator, x, importance)   methods/nestedlora.py:254-267
 *     loss.backward(); optimizer.step(); ema.update()                     examples/operator/__init__.py:55-74
 *     compute_spectrum_evd(...)                                           methods/spectrum.py:29-102
 *
 * The reference has no FFI (it is pure Python on torch eager ops); the entry points below are the
 * functions a maintainer would bind (ctypes stub in INTEGRATION.md) to replace the torch op
 * sequences cited per function.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous float32 unless it says "host";
 *   - the caller owns every buffer, including the workspace (size: nsvd_workspace_bytes);
 *     nothing here allocates, frees or synchronises;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), is safe to capture in
 *     a HIP graph (no allocation, no synchronisation, no host read-back: tests/test_graph_gpu.py captures and
 *     replays the training step), and keeps no global mutable state (re-entrant across streams / devices).
 *     The scalars that change from step to step (scheduled learning rate, EMA decay, sampler counter) are kernel
 *     ARGUMENTS in the plain entry points - a captured step replays them frozen - and live in device memory
 *     when the caller passes a nsvd_step_state (below): that form replays a moving schedule;
 *   - return value: 0 on success, negative on error (-(int)hipError_t for HIP failures,
 *     NSVD_EINVAL / NSVD_EUNSUPPORTED for argument errors); no C++ exception crosses the ABI.
 *
 * Tensor layouts (B = batch rows, D = space dims, E = 1 + 2D stencil points, R = E*B stencil
 * rows ordered [x, x+eps e_0, x-eps e_0, x+eps e_1, ...] x B, L = eigenfunctions/heads,
 * m = Fourier mapping size, F = 2m features, h_i = width of layer i, h_last = 1):
 *     x        (B, D)           row-major
 *     fourier_B(D, m)           examples/utils.py:116-121  (`_B`, frozen)
 *     W_i      (L, h_i, h_{i-1})examples/models/mlp.py:187  (`ws[i]`)
 *     b_i      (L, h_i, 1)      examples/models/mlp.py:189  (`bs[i]`)
 *     scales   (L,)             examples/operator/pde/boundary.py:43 (ExponentialMask), may be NULL
 *     f, Tf    (B, L)           what operator(model, x, importance) returns
 *     lam      (2, L, L)        lam_f1, lam_f2 of methods/nestedlora.py:89
 */
#ifndef NSVD_H
#define NSVD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSVD_ABI_VERSION 2
#define NSVD_MAX_LAYERS 8

#define NSVD_EINVAL (-10001)
#define NSVD_EUNSUPPORTED (-10002)

/* potentials: examples/operator/pde/schrodinger/potentials.py:5-8 and :24-27 */
#define NSVD_POT_HYDROGEN 0 /* V = -Z / |x|  */
#define NSVD_POT_HARMONIC 1 /* V = k |x|^2   */

/* nesting masks: methods/nestedlora.py:40-54 */
#define NSVD_MASK_CUSTOM 0     /* v (L) and M (L,L) given by pointer                    */
#define NSVD_MASK_SEQUENTIAL 1 /* v = 1, M = triu(1): generated in registers, v/M NULL  */
#define NSVD_MASK_JOINT 2      /* step 1: v_l = (L-l)/L, M = min(v_i, v_j); v/M NULL    */

/* which implementation nsvd_operator_* picks */
#define NSVD_PATH_AUTO 0    /* fused MFMA kernels when the shape allows, else generic   */
#define NSVD_PATH_GENERIC 1 /* layer-by-layer generic kernels (any shape)               */
#define NSVD_PATH_FUSED 2   /* fused MFMA kernels or NSVD_EUNSUPPORTED                  */
#define NSVD_PATH_FUSED_BF16X3 3 /* opt-in, never picked by AUTO: the fused forward with every layer on the bf16 MFMA,
                                  * every float32 operand split into three bf16 planes and six partial products
                                  * accumulated in float32 (f and Tf as close to float64 as the fp32 MFMA's). The
                                  * backward is NSVD_PATH_FUSED's; a fused training step on this path also leaves
                                  * the planes of the updated weights for the next forward (nsvd_step_emits_planes) */

/* Shape of WaveFunctions(ParallelMLP(GaussianFourierFeatureTransform)):
 * examples/operator/pde/__init__.py:19-55, examples/models/mlp.py:167-221, examples/utils.py:90-143 */
typedef struct nsvd_model_desc {
    int32_t L;                       /* --neigs                                        */
    int32_t D;                       /* --ndim * n_particles                           */
    int32_t m;                       /* --fourier_mapping_size                         */
    int32_t nlayers;                 /* number of weight matrices = len(hidden) + 1    */
    int32_t dims[NSVD_MAX_LAYERS];   /* h_0 .. h_{nlayers-1}; the last one must be 1   */
    int32_t has_exp_mask;            /* --apply_exp_mask                               */
} nsvd_model_desc;

/* Device pointers of one parameter set (weights, their gradients, RMSprop state or EMA shadow
 * all use this same struct). */
typedef struct nsvd_params {
    float* fourier_B;                /* (D, m); unused in gradient/state sets          */
    float* W[NSVD_MAX_LAYERS];
    float* b[NSVD_MAX_LAYERS];
    float* scales;                   /* NULL when has_exp_mask == 0                    */
} nsvd_params;

/* OperatorWrapper(NegativeHamiltonian(potential), scale, shift) with Gaussian importance:
 * examples/__init__.py:1-9, examples/operator/pde/schrodinger/__init__.py:4-22,
 * examples/operator/pde/diff_ops.py:4-52, examples/operator/pde/main_pde.py:94-100 */
typedef struct nsvd_problem {
    int32_t potential;               /* NSVD_POT_*                                     */
    float charge_or_k;               /* Z (hydrogen) or k (oscillator)                 */
    float scale_kinetic;             /* problems.py:26 (1.0)                           */
    float eps;                       /* --laplacian_eps: > 0 central differences; <= 0 the exact Laplacian
                                      * (diff_ops.py:7,54-61) by forward-mode jets, MFMA path only */
    float op_scale;                  /* --operator_scale                               */
    float op_shift;                  /* --operator_shift                               */
    float sigma;                     /* --sampling_scale of the Gaussian sampler       */
    float hard_mul_const;            /* --hard_mul_const                               */
    int32_t use_importance;          /* 1: importance = N(0, sigma^2 I) pdf; 0: None   */
} nsvd_problem;

int nsvd_abi_version(void);

/* Name of the implementation nsvd_operator_forward would take ("fused_mfma", "generic"). host */
const char* nsvd_path_name(const nsvd_model_desc* desc, int B, int path);
/* The same for a given problem: the exact-Laplacian mode (prob->eps <= 0) exists on the MFMA path only (D <= 3, as the
 * stencil mode: its 3-D form runs one direction per workgroup); "unsupported" when it has no path. */
const char* nsvd_path_name_for(const nsvd_model_desc* desc, const nsvd_problem* prob, int B, int path);

/* Bytes of scratch nsvd_operator_forward / _backward need for batches of up to B rows. */
size_t nsvd_workspace_bytes(const nsvd_model_desc* desc, int B);

/* phiT[(j | m + j), r] = (sin | cos)(x_r . B_j) for the R = E*B stencil rows of x.
 * Replaces GaussianFourierFeatureTransform.forward at the 1+2D points of
 * VectorizedLaplacian.approx_laplacian (examples/utils.py:126-143, diff_ops.py:36-45).
 * phiT: (2m, ldr) feature-major, sample-contiguous; ldr >= E*B. nstencil = 1 (centre only) or E. */
int nsvd_fourier_features(const float* x, const float* fourier_B, float* phiT, int B, int D, int m,
                          float eps, int nstencil, int ldr, void* stream);

/* bit for nsvd_operator_forward's save_for_backward argument: the Fourier features of x are already in
 * the workspace (put there by nsvd_operator_features, typically on another stream while the previous
 * step's optimiser runs - they do not depend on the trainable weights) */
#define NSVD_FEATURES_READY 0x100
/* bit for nsvd_operator_forward's save_for_backward argument, path NSVD_PATH_FUSED_BF16X3 only: the bfloat16 planes of
 * the CURRENT weights are already in this workspace - left there by the nsvd_operator_backward_evd_step[_next] call
 * (same path) that produced these weights, when nsvd_step_emits_planes says it does: the forward then needs no split
 * launch (9 us of 167 at the headline configuration). The caller answers for "current": any other change of the
 * parameters between that step and this forward (a load, another optimiser) invalidates the planes. */
#define NSVD_W_PLANES_READY 0x200
/* Does a fused training step on this shape and path write the planes of the updated weights (1) or not (0)?
 * (NSVD_PATH_FUSED_BF16X3 on a shape the MFMA kernels take, weight gradients without batch slices.) They go into the
 * workspace the next forward reads: `next_ws` of nsvd_operator_backward_evd_step_next, `ws` of .._step. */
int nsvd_step_emits_planes(const nsvd_model_desc* desc, int B, int path);

/* The weight-independent prologue of nsvd_operator_forward alone: Fourier features of the stencil rows of
 * x into `ws`, in the layout the path selected by (desc, B, path) reads. */
int nsvd_operator_features(const nsvd_model_desc* desc, const nsvd_params* params, const nsvd_problem* prob,
                           const float* x, int B, void* ws, size_t ws_bytes, int save_for_backward, int path,
                           void* stream);

/* Tf, f = operator(method, x, importance):  the 1+2D ParallelMLP evaluations, importance
 * re-weighting, central-difference Laplacian, potential, scale/shift
 * (examples/__init__.py:7-9 -> schrodinger/__init__.py:16-22 -> diff_ops.py:9-52 ->
 *  pde/__init__.py:15-16 -> models/mlp.py:204-221).
 * The stencil is the reference's - the same 1 + 2 D points, the same eps - but every path carries the shifted
 * evaluations as EVEN / ODD perturbations of the centre one (z(x +- eps e_d) = z + zE_d +- zO_d through the Fourier map,
 * every layer and the re-weighting), so the float32 result is the stencil's value to ~1e-6 instead of
 * the few per cent a point-wise float32 difference at eps = 0.01 carries (the reference's own float32 Tf is 4e-2 from
 * its float64 Tf: BASELINE.md). f is unaffected. Any eps > 0 (perturbations beyond 0.25 fall back to plain differences).
 * save_for_backward != 0 keeps the centre-row pre-activations in `ws` for nsvd_operator_backward. */
int nsvd_operator_forward(const nsvd_model_desc* desc, const nsvd_params* params,
                          const nsvd_problem* prob, const float* x, int B, float* f, float* Tf,
                          void* ws, size_t ws_bytes, int save_for_backward, int path, void* stream);

/* Parameter gradients of sum(df * f) (f is the only differentiable output: Tf receives no
 * gradient, methods/nestedlora.py:108-111).  Replaces autograd through the centre evaluation.
 * `ws` must be the workspace the matching nsvd_operator_forward(save_for_backward=1) filled.
 * Gradients are OVERWRITTEN (the reference calls optimizer.zero_grad() every step). */
int nsvd_operator_backward(const nsvd_model_desc* desc, const nsvd_params* params,
                           const nsvd_problem* prob, const float* x, int B, const float* df,
                           const nsvd_params* grads, void* ws, size_t ws_bytes, int path, void* stream);

/* out[b, l] = hard_mul_const * base_l(x_b) * mask_l(x_b): WaveFunctions.forward, i.e. what
 * NestedLoRA.forward / method(x) returns (examples/operator/pde/__init__.py:15-16,
 * methods/nestedlora.py:195-200); out: (B, L). save_for_backward != 0 keeps what nsvd_model_backward needs
 * in `ws` (used by compute_loss_kernel-style callers that differentiate through model(x) itself). */
int nsvd_model_forward(const nsvd_model_desc* desc, const nsvd_params* params, const float* x, int B,
                       float hard_mul_const, float* out, void* ws, size_t ws_bytes, int save_for_backward,
                       void* stream);

/* Draws the batch AND prepares its features in one launch: x[b][d] = sigma * N(0,1) with sigma = prob->sigma,
 * from a counter-based generator (Philox4x32-10 + Box-Muller) keyed by (seed, offset, b): the device-side form of
 * `x = sampling_scale * torch.randn(batch_size, ndim)` + host->device copy (examples/operator/pde/main_pde.py:92-93,
 * examples/operator/__init__.py:58). x (B, D) is an OUTPUT; follow with nsvd_operator_forward(... |
 * NSVD_FEATURES_READY). The stream of values is a pure function of (seed, offset): pass a fresh offset per batch;
 * ranks that pass the same (seed, offset) draw the same batch. Not torch's generator: same distribution, different
 * numbers. */
int nsvd_operator_sample_features(const nsvd_model_desc* desc, const nsvd_params* params, const nsvd_problem* prob,
                                  unsigned long long seed, unsigned long long offset, float* x, int B, void* ws,
                                  size_t ws_bytes, int save_for_backward, int path, void* stream);

/* Workspace of nsvd_model_forward / nsvd_model_backward alone (no stencil rows): smaller than
 * nsvd_workspace_bytes, and defined for input dimensions up to 64 (the operator entry points stop at D = 4). */
size_t nsvd_model_workspace_bytes(const nsvd_model_desc* desc, int B);

/* Parameter gradients of sum(dout * model(x)) for the matching nsvd_model_forward(save_for_backward=1). */
int nsvd_model_backward(const nsvd_model_desc* desc, const nsvd_params* params, const float* x, int B,
                        const float* dout, const nsvd_params* grads, void* ws, size_t ws_bytes, void* stream);

/* NestedLoRALossFunctionEVD.forward with f1, f2 = chunk(f, 2) (methods/nestedlora.py:70-94, :263).
 * Stage 1: moments[0 : L*L] = lam_f1, [L*L : 2*L*L] = lam_f2, [2*L*L] = mean_b sum_l v_l f Tf.
 * These 2L^2+1 floats are the data-parallel exchange payload (all-reduce mean over ranks).
 * scratch: nsvd_evd_scratch_bytes(B, L) bytes. */
size_t nsvd_evd_scratch_bytes(int B, int L);
int nsvd_evd_moments(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                     float* moments, void* scratch, void* stream);

/* Stage 2: loss[0] = -2 * moments[2L^2] + sum(M * lam_f1 * lam_f2); loss[1] = operator term,
 * loss[2] = metric term; and, when df != NULL, NestedLoRALossFunctionEVD.backward
 * (methods/nestedlora.py:98-111) with the f1/f2 contributions summed into d loss / d f:
 *   df[b, :] = grad_scale * ( -(4/B) v * Tf[b] + (2/B_half) f[b] @ (M * lam_other) ).
 * grad_scale folds grad_output and, for data parallel runs, nothing else (ranks average later). */
int nsvd_evd_loss_grad(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                       const float* M, const float* moments, float grad_scale, float* loss, float* df,
                       void* stream);

/* Stages 1 + 2 in one call for the single-GPU case (no exchange between them): one launch when
 * B*L <= 16384, the two-stage pipeline otherwise. Same outputs as the two calls above. */
int nsvd_evd_loss_fused(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v,
                        const float* M, float grad_scale, float* moments, float* loss, float* df,
                        void* scratch, void* stream);

/* Stage 1a alone: the per-chunk partial sums of the moments into `scratch` (nsvd_evd_scratch_bytes),
 * without the reduction launch. Consumed by nsvd_operator_backward_evd(moments_reduced = 0). */
int nsvd_evd_partial(const float* f, const float* Tf, int B, int L, int mask_kind, const float* v, void* scratch,
                     void* stream);

/* Heads sharded over `world` processes: gathered = the ranks' packed blocks as an all-gather leaves them,
 * (world, 2, B, L_local) = [rank][f | Tf][row][local head]; writes the (B, world * L_local) arrays f and Tf every
 * other entry point reads and - scratch != NULL - the per-chunk partial moments exactly as nsvd_evd_partial(f, Tf)
 * would (same bits), in the same launch. (The reference has no sharded form: methods/nestedlora.py:70-94 sees the
 * whole (B, L) f.) */
int nsvd_evd_gather_heads(const float* gathered, int world, int B, int L_local, int mask_kind, const float* v,
                          float* f, float* Tf, void* scratch, void* stream);

/* The same for ANY head count L >= world (the reference scripts run --neigs 36 / 55, scripts/exps/pde/hydrogen.sh:28,
 * oscillator.sh:27): rank w owns n_w = L / world + (w < L % world) consecutive heads starting at
 * w (L / world) + min(w, L % world) - the first L % world ranks one head more. `gathered` holds `world` blocks of
 * 2 B ceil(L / world) floats (all-gather blocks are equally long); block w begins with f (B, n_w) then Tf (B, n_w),
 * packed, the rest of a short block is never read. With L % world == 0 this IS nsvd_evd_gather_heads. */
int nsvd_evd_gather_head_blocks(const float* gathered, int world, int B, int L, int mask_kind, const float* v,
                                float* f, float* Tf, void* scratch, void* stream);

/* nsvd_evd_loss_grad + nsvd_operator_backward in ONE call: d loss / d f (methods/nestedlora.py:98-111) is
 * evaluated per sample inside the backward kernels and never stored. `moments` (2L^2+1 floats) is
 *   - an INPUT when moments_reduced != 0 (e.g. after the data-parallel all-reduce of nsvd_evd_moments), or
 *   - an OUTPUT when moments_reduced == 0: the partial sums in `evd_scratch` (from nsvd_evd_partial on the
 *     same f, Tf) are reduced on the fly and the reduced vector is stored here.
 *   - ignored (may be NULL) when moments_reduced == 0 and evd_scratch == NULL ("direct" form, MFMA path only):
 *     every workgroup of the backward takes the 2 L moments of its own head from f itself, no moment kernel
 *     runs at all and `moments` is not written. `loss` (when not NULL, all heads local, one head window) IS: the
 *     workgroups leave per-head partial sums of the two loss terms and the weight-gradient kernel adds them in head
 *     order - the loss value of every step (the reference's loss.item(), examples/operator/__init__.py:74) at no launch.
 * loss[0..2] = {loss, operator term, metric term}. Gradients are overwritten as in nsvd_operator_backward.
 * Head-parallel sharding: f, Tf, v, M and the moments may cover L_total >= desc->L heads, of which this
 * model owns [l_offset, l_offset + desc->L) (f, Tf are then (B, L_total), gathered from all ranks); pass
 * L_total = 0 / l_offset = 0 otherwise. */
int nsvd_operator_backward_evd(const nsvd_model_desc* desc, const nsvd_params* params,
                               const nsvd_problem* prob, const float* x, int B, const float* f, const float* Tf,
                               int mask_kind, const float* v, const float* M, float* moments, int moments_reduced,
                               const void* evd_scratch, int L_total, int l_offset, float grad_scale, float* loss,
                               const nsvd_params* grads, void* ws, size_t ws_bytes, int path, void* stream);

/* nsvd_operator_backward_evd for a WINDOW of heads: only the gradients of heads [l_begin, l_begin + l_count) of
 * this model are produced (all other gradient elements are left untouched). The L heads of ParallelMLP share nothing
 * but the input (examples/models/mlp.py:187-189, 204-221), so autograd's backward of the step
 * (examples/operator/__init__.py:68) splits exactly by head; a sample-sharded run calls this once per window and
 * starts the all-reduce of a window's gradients while the next window is being computed. Calling it for every window
 * of a partition of [0, L) gives bit for bit the gradients of one nsvd_operator_backward_evd call of the same tile
 * shape. Every call must be handed the moments (moments_reduced != 0, or the same evd_scratch): the windows of one
 * step share f, Tf and the workspace of the forward call.
 * nsvd_backward_head_window_ok: 1 when windows of l_count heads are supported for this shape (MFMA path, no split-K
 * partial buffers at that head count), else 0 - the caller then uses one window of all L heads. */
int nsvd_operator_backward_evd_heads(const nsvd_model_desc* desc, const nsvd_params* params,
                                     const nsvd_problem* prob, const float* x, int B, const float* f,
                                     const float* Tf, int mask_kind, const float* v, const float* M, float* moments,
                                     int moments_reduced, const void* evd_scratch, int L_total, int l_offset,
                                     float grad_scale, float* loss, const nsvd_params* grads, void* ws,
                                     size_t ws_bytes, int path, int l_begin, int l_count, void* stream);
int nsvd_backward_head_window_ok(const nsvd_model_desc* desc, const nsvd_problem* prob, int B, int path,
                                 int l_count);

/* nsvd_operator_backward_evd with the optimiser step of nsvd_rmsprop_ema_step taken inside the
 * weight-gradient kernel: each gradient element updates its parameter (params->W/b/scales, IN PLACE), RMSprop
 * square average (opt->sq) and EMA shadow (opt->ema, when has_ema) as it leaves the accumulator, so gradients
 * never make the HBM round trip and the optimiser launch disappears (optimizer.step() + ema.update() of
 * examples/operator/__init__.py:69-73 folded into loss.backward() of :68). Bit-identical to the two separate
 * calls. grads may be NULL on the fused path (gradients are then not stored at all); on the generic path grads
 * is required and the step is taken by per-tensor optimiser launches. lr / ema_decay are the already scheduled
 * values, as for nsvd_rmsprop_ema_step. Not for data-parallel runs (gradients must be all-reduced first). */
typedef struct nsvd_step_state nsvd_step_state;  /* device-resident schedule state, declared below */
typedef struct nsvd_rmsprop {
    nsvd_params sq;                  /* RMSprop square averages, parameter layouts      */
    nsvd_params ema;                 /* EMA shadow parameters (ignored unless has_ema)  */
    double lr, alpha, eps, ema_decay;
    int32_t has_ema;
    /* NULL: lr / ema_decay above are the already scheduled values of this step (kernel arguments).
     * Non-NULL (DEVICE pointer, nsvd_step_state_init): lr, alpha, eps, ema_decay above are ignored; the first
     * backward kernel of the step derives the step's values from state->step, the optimiser epilogue reads them from
     * the device, and the last kernel of the step increments state->step - the call can be captured in a HIP graph
     * and replayed along the schedule. Fused MFMA path only (NSVD_EUNSUPPORTED otherwise). */
    nsvd_step_state* state;
} nsvd_rmsprop;
int nsvd_operator_backward_evd_step(const nsvd_model_desc* desc, const nsvd_params* params,
                                    const nsvd_problem* prob, const float* x, int B, const float* f,
                                    const float* Tf, int mask_kind, const float* v, const float* M,
                                    float* moments, int moments_reduced, const void* evd_scratch, int L_total,
                                    int l_offset, float grad_scale, float* loss, const nsvd_params* grads,
                                    const nsvd_rmsprop* opt, void* ws, size_t ws_bytes, int path, void* stream);

/* ---- device-resident schedule state --------------------------------------------------------------------------
 * What changes from one iteration of the loop body examples/operator/__init__.py:55-74 to the next besides the
 * tensors: CosineAnnealingLR's learning rate (:35,71-72; closed form eta_min + (lr0 - eta_min)(1 + cos(pi t / T)) / 2
 * after t scheduler steps), torch_ema's warmed-up decay min(decay, (1 + n) / (10 + n)) at its n-th update (:36,73),
 * and - for the device sampler - the batch counter. A training step whose kernels take these as launch arguments
 * cannot be replayed from a captured HIP graph. This struct lives in DEVICE memory (the caller allocates
 * sizeof(nsvd_step_state) bytes, 8-byte aligned, and fills it with nsvd_step_state_init); kernels read the step's
 * values from `cur` and the last kernel of a step increments `step`.
 * Arithmetic: the same double-precision expressions the host path evaluates (Python floats), rounded to float32
 * where torch rounds; the device cosine may differ from libm's in the last bit of the DOUBLE, i.e. the float32
 * learning rate agrees with the host path's except with probability ~2^-29 per step. */
struct nsvd_step_state {
    uint64_t step;                   /* optimiser steps taken = scheduler steps = torch_ema.num_updates          */
    uint64_t T_max;                  /* CosineAnnealingLR T_max (--num_iters); 0: constant learning rate          */
    double lr0, eta_min;             /* base learning rate (--lr), CosineAnnealingLR eta_min (0 in the scripts)    */
    double alpha, eps;               /* RMSprop alpha (--rmsprop_decay), eps (1e-10: examples/utils.py:52)         */
    double ema_decay;                /* --ema_decay; the warm-up of torch_ema (use_num_updates=True) is applied    */
    /* the values of the step being taken, derived from `step` by the step's first kernel (or nsvd_step_state_begin): */
    struct {
        float lr, alpha, one_minus_alpha, eps, one_minus_decay, grad_scale;
    } cur;
    uint64_t reserved;
};
/* Fill a device-resident state (one tiny launch on `stream`): counters at `step`, `cur` = the values of step `step`. */
int nsvd_step_state_init(nsvd_step_state* state, double lr0, double eta_min, unsigned long long T_max, double alpha,
                         double eps, double ema_decay, unsigned long long step, void* stream);
/* state->cur <- the values of step state->step (one tiny launch). For loop bodies whose first kernel does not do it:
 * i.e. everything except nsvd_operator_backward_evd_step[_next] with opt->state set. */
int nsvd_step_state_begin(nsvd_step_state* state, void* stream);
/* nsvd_rmsprop_ema_step with the scheduled values read from state->cur (optimizer.step() + scheduler.step() +
 * ema.update() of examples/operator/__init__.py:69-73 in capturable form); advance != 0: this is the last optimiser
 * launch of the step - one thread increments state->step when the kernel is done with it. */
int nsvd_rmsprop_ema_step_dev(float* p, const float* grad, float* sq, float* ema, size_t n, nsvd_step_state* state,
                              double grad_scale, int advance, void* stream);
/* nsvd_operator_sample_features whose batch counter is offset_base + state->step, read on the device: the draw of a
 * captured step moves along with the schedule. */
int nsvd_operator_sample_features_dev(const nsvd_model_desc* desc, const nsvd_params* params,
                                      const nsvd_problem* prob, unsigned long long seed,
                                      unsigned long long offset_base, const nsvd_step_state* state, float* x, int B,
                                      void* ws, size_t ws_bytes, int save_for_backward, int path, void* stream);

/* torch.optim.RMSprop(alpha, eps, momentum=0, centered=False) step + torch_ema update, fused
 * (examples/utils.py:50-57, examples/operator/__init__.py:69-73):
 *   g = grad_scale * grad; sq = alpha sq + (1-alpha) g^2; p -= lr g / (sqrt(sq) + eps);
 *   ema -= (1 - ema_decay) (ema - p)        (skipped when ema == NULL)
 * over n contiguous floats. lr is the already-scheduled learning rate, ema_decay the already
 * warmed-up decay min(decay, (1+t)/(10+t)). Host scalars are doubles (Python floats) and are rounded
 * to float32 exactly where torch rounds them: alpha, (1 - alpha), lr, eps, (1 - ema_decay). */
int nsvd_rmsprop_ema_step(float* p, const float* grad, float* sq, float* ema, size_t n, double lr,
                          double alpha, double eps, double ema_decay, double grad_scale, void* stream);

/* compute_spectrum_evd accumulation for one chunk (methods/spectrum.py:56-75):
 *   w = sqrt(p_train(x)) / sqrt(p_val), phi = nan_to_num(w f), Tphi = nan_to_num(w Tf),
 *   Tphi rows with x ~ 0 zeroed, cov += phi^T phi, quad += phi^T Tphi   (cov, quad: (L, L)).
 * p_val = uniform on [-lim, lim]^D (main_pde.py:129-130). */
int nsvd_spectrum_accumulate(const float* f, const float* Tf, const float* x, int B, int L, int D,
                             float sigma, int use_importance, float lim, float* cov, float* quad,
                             void* stream);
/* The same accumulation with float64 products, sums and accumulators (cov, quad: (L, L) doubles). The reference keeps
 * float32 accumulators (methods/spectrum.py:60-61,74-75); for excited states diag(quad) is a sum of terms up to 10^2 x
 * larger than itself, which a float32 running sum carries to ~1e-4 only - the evaluation paths of this package
 * (trainer.spectrum, spectrum.compute_spectrum_evd) accumulate here and round once at the end. */
int nsvd_spectrum_accumulate_f64(const float* f, const float* Tf, const float* x, int B, int L, int D,
                                 float sigma, int use_importance, float lim, double* cov, double* quad,
                                 void* stream);
/* compute_spectrum_evd(set_first_mode_const=True) (methods/spectrum.py:68-70): the same float64 accumulation with a
 * constant-one column padded in FRONT of the weighted phi and Tphi (after the weighting, before nan_to_num and the
 * x ~ 0 zeroing of Tphi); cov, quad: (L + 1, L + 1) doubles. */
int nsvd_spectrum_accumulate_const_f64(const float* f, const float* Tf, const float* x, int B, int L, int D,
                                       float sigma, int use_importance, float lim, double* cov, double* quad,
                                       void* stream);

/* ---- next row: dense kernel operator on a minibatch (kernel-operator configuration) ---------------------------
 * Kf[i][l] = scale * sum_k K[rows[i]][cols[k]] f[k][l]: the (Kf, f) producer that
 * NestedLoRA.compute_loss_kernel's `get_approx_kernel_op(x)(model, x, importance)` contract consumes
 * (methods/nestedlora.py:230-252; split_batch: rows = x1, cols = x2, f = model(x2), scale = 1 / B2). The reference
 * ships no kernel operator; the definition is this build's (K = A A^T / r + 1e-3 I on N points,
 * minibatch of indices drawn with replacement), restated in float64 by oracle/nsvd_oracle.py:kernel_apply.
 * K: (N, ldk) row-major float32, ldk >= N rounded up to 64 (rows are read in whole 64-float chunks; the padding
 * may hold anything finite), 16-byte aligned; rows (B1), cols (B2): int64 point indices; f: (B2, L); out: (B1, L).
 * Indices outside [0, N) contribute / produce zeros. */
size_t nsvd_kernel_apply_workspace_bytes(int N, int B1, int L);
int nsvd_kernel_apply(const float* K, size_t ldk, int N, const long long* rows, int B1, const long long* cols,
                      int B2, const float* f, int L, float scale, float* out, void* ws, size_t ws_bytes,
                      void* stream);

/* ---- next row: the CDK (two-tower) NestedLoRA loss ------------------------------------------------------
 * NestedLoRALossFunctionForCDK (methods/nestedlora.py:273-332) as called by NestedLoRAForCDK.compute_loss
 * (methods/nestedlora.py:366-378; examples/cdk/sketchy/main_sketchy.py:188).
 * f, g: (B, L) row-major float32 tower outputs. With Lp = L + (set_first_mode_const != 0):
 *   f~ = batch_weights[:, None] * [1, f]  (constant first mode when set_first_mode_const; weights optional, (B,))
 *   lam_f = f~^T f~ / B, lam_g = g~^T g~ / B
 *   loss[0] = loss[1] + loss[2]; loss[1] = -2 mean_b sum_l v_l f~ g~; loss[2] = sum(M * lam_f * lam_g)
 *   rs_joint (B) = diag(f~ g~^T), rs_indep (B (B-1)) = off_diagonal(f~ g~^T) (methods/utils.py:16-22); either
 *   may be NULL (both NULL skips the (B, B) gram contraction).
 * v: (Lp), M: (Lp, Lp) nesting masks (methods/nestedlora.py:345-359), float32 on the device.
 * ws: nsvd_cdk_workspace_bytes(B, L, set_first_mode_const) bytes; keeps f~, g~ and M * lam for the backward.
 * Arithmetic is float32 on the fp32-input MFMA, independent of any autocast state of the caller. */
size_t nsvd_cdk_workspace_bytes(int B, int L, int set_first_mode_const);
int nsvd_cdk_loss_forward(const float* f, const float* g, const float* batch_weights, const float* v,
                          const float* M, int B, int L, int set_first_mode_const, float* loss, float* rs_joint,
                          float* rs_indep, void* ws, size_t ws_bytes, void* stream);

/* NestedLoRALossFunctionForCDK.backward (methods/nestedlora.py:309-332) after nsvd_cdk_loss_forward on the
 * same ws:  grad_f = grad_out * ( -(2/B) g~ v + (2/B) f~ (M * lam_g) )[:, first:], grad_g with f <-> g.
 * As in the reference the result is the gradient w.r.t. the WEIGHTED features (batch_weights are not
 * chain-ruled) and the constant column is dropped. grad_out: device scalar (the autograd grad_output, e.g.
 * the AMP loss scale) or NULL for 1. grad_f / grad_g: (B, L), either may be NULL. */
int nsvd_cdk_loss_backward(const float* v, int B, int L, int set_first_mode_const, const float* grad_out,
                           float* grad_f, float* grad_g, void* ws, size_t ws_bytes, void* stream);

/* normalize(z, r_up, regularize_mode) applied to the CDK towers' embeddings (examples/models/siam.py:170-183, called
 * from HeteroNetwork.forward_single :156-166). z, out, dout, dz: (B, L) float32 row-major.
 *   NSVD_NORMALIZE_L2_BALL:   rows with ||z|| < r_up unchanged, the others r_up * z / max(||z||, 1e-12)
 *   NSVD_NORMALIZE_L2_SPHERE: every row r_up * z / max(||z||, 1e-12)
 * Backward of the same expression (the comparison is a constant mask, as in the reference). out may alias z in the
 * forward; dz may alias dout in the backward. */
#define NSVD_NORMALIZE_L2_BALL 0
#define NSVD_NORMALIZE_L2_SPHERE 1
int nsvd_row_normalize_forward(const float* z, int B, int L, float r_up, int mode, float* out, void* stream);
int nsvd_row_normalize_backward(const float* z, const float* dout, int B, int L, float r_up, int mode, float* dz,
                                void* stream);

/* One CDK tower: Linear(d0 -> d1) -> BatchNorm1d(d1) -> LeakyReLU(slope) -> Linear(d1 -> d2) -> BatchNorm1d(d2), what
 * get_mlp(sizes=[d0, d1, d2], bias=True, nonlinearity='lrelu<slope>', use_bn=True) builds (examples/models/mlp.py:129-164)
 * and main_sketchy.py:107-116 uses as the two backbones of HeteroNetwork (examples/models/siam.py:132-166), in
 * TRAINING mode (batch statistics; running_mean / running_var updated with `momentum` when update_running != 0, as
 * torch.nn.BatchNorm1d does). Replaces the module's forward and its autograd backward. fp32 MFMA contractions.
 *   x (B, d0); z (B, d2); dz (B, d2); every parameter / gradient in torch's layout: W1 (d1, d0), b1 (d1), g1 / be1 /
 *   rm1 / rv1 (d1) = BatchNorm weight / bias / running_mean / running_var, W2 (d2, d1), b2, g2, be2, rm2, rv2 (d2).
 *   Shapes: B, d0, d1, d2 multiples of 128, B <= 1024 (anything else: NSVD_EINVAL; 0 from the size query).
 *   The forward leaves what the backward needs in `ws` (nsvd_tower_workspace_bytes): call the backward with the same
 *   x, parameters and workspace. The gradient w.r.t. x is not produced (the towers' inputs are data). */
typedef struct nsvd_tower_params {
    float *W1, *b1, *g1, *be1, *rm1, *rv1;
    float *W2, *b2, *g2, *be2, *rm2, *rv2;
} nsvd_tower_params;
size_t nsvd_tower_workspace_bytes(int B, int d0, int d1, int d2);
/*   gemm_bf16 != 0: MIXED PRECISION, the role of the reference's autocast branch (examples/cdk/sketchy/
 *   main_sketchy.py:161,182, on by default there: Linear and BatchNorm / activation outputs are half tensors, statistics
 *   and master weights float32) with bfloat16 as the half type and no GradScaler: X, W1, W2 are rounded to bfloat16
 *   (round to nearest even), the wide activations Y1 = X W1^T + b1, A1 = lrelu(BN1(Y1)) and the gradients dY2, dA1, dY1
 *   are STORED as bfloat16, the five contractions run on the bf16 MFMA with float32 accumulation (csrc/gemm16.h, no
 *   transposed copies), BatchNorm statistics, the narrow end (Y2, Z) and every parameter gradient stay float32.
 *   Not bit-comparable with float16 autocast; pinned to the float64 oracle with the same roundings. Shapes:
 *   nsvd_tower_mixed_supported (B, d1, d2 multiples of 256, d0 of 128, B <= 1024), NSVD_EUNSUPPORTED otherwise. The
 *   backward must be called with the flag of its forward (the workspace holds bfloat16 activations).
 *   Bit 1 (value 2) of the flag, forward only: the bfloat16 copies of W1 / W2 inside `ws` are current - set by callers
 *   that know the weights have not changed since the copies were written (nsvd_cdk_step's optimiser kernel refreshes
 *   them while it updates the float32 masters); clear, the forward casts the masters first. */
int nsvd_tower_mixed_supported(int B, int d0, int d1, int d2);
/*   Which of the two mixed-precision forms a call with this shape and activation slope runs. 1: the wide layer with
 *   BatchNorm INSIDE the contraction (csrc/tower_col.h, slope > 0): Y1 and dA1 stay float32 in the accumulators and are
 *   never stored (statistics, normalisation and the activation on unrounded values); the backward recovers the normalised
 *   value from the stored bfloat16 activation, h = A1 > 0 ? A1 : A1 / slope, yhat = (h - beta1) / gamma1 (needs
 *   gamma1 != 0). 0: contraction + strip kernels, Y1 and dA1 stored as bfloat16 as described above (slope == 0: ReLU is
 *   not invertible). The float64 oracle restates both (oracle.tower_forward_backward: gemm_bf16 = "fused" / True). */
int nsvd_tower_mixed_fused(int B, int d0, int d1, int d2, float slope);
int nsvd_tower_forward(const float* x, const nsvd_tower_params* params, int B, int d0, int d1, int d2, float slope,
                       float eps, float momentum, int update_running, int gemm_bf16, float* z, void* ws,
                       size_t ws_bytes, void* stream);
int nsvd_tower_backward(const float* x, const nsvd_tower_params* params, const float* dz, int B, int d0, int d1,
                        int d2, float slope, int gemm_bf16, const nsvd_tower_params* grads, void* ws, size_t ws_bytes,
                        void* stream);
/* A tower whose HIDDEN width is sharded over processes (rank r holds rows [r d1/W, (r+1) d1/W) of W1, b1 and the
 * first BatchNorm, and the same columns of W2; b2 and the second BatchNorm are replicated): d1 is the LOCAL width.
 * BatchNorm statistics are per column, so the first half needs no exchange and the arithmetic is the unsharded tower's
 * (the reference's is single-process: examples/models/mlp.py:129-164). phase 1: Linear1, BatchNorm1, activation and this
 * rank's PARTIAL product A1 W2^T (no bias) into the workspace at nsvd_tower_y2_offset - (B, d2) floats the caller sums
 * over the ranks in place (one all-reduce); phase 2: + b2, BatchNorm2 -> z. phase 0 = nsvd_tower_forward. The
 * backward (nsvd_tower_backward with the local d1) needs no exchange: dz is replicated, every other quantity local. */
size_t nsvd_tower_y2_offset(int B, int d0, int d1, int d2);
int nsvd_tower_forward_phase(const float* x, const nsvd_tower_params* params, int B, int d0, int d1, int d2,
                             float slope, float eps, float momentum, int update_running, int gemm_bf16, int phase,
                             float* z, void* ws, size_t ws_bytes, void* stream);

/* nsvd_operator_backward_evd_step for the heads [l_begin, l_begin + l_count) only: ONE fused step taken as several head
 * windows (the heads of ParallelMLP share nothing but the input, examples/models/mlp.py:187-189), each window its own
 * pair of launches, possibly on its own stream - a two-window step lets the latency-bound chain of window 1 and the
 * HBM-bound optimiser epilogue of window 0 sit under the other window's MFMA loops (trainer.FusedTrainer
 * (backward_windows=2)). Every window of a step gets the same arguments except l_begin / l_count, last_window and
 * ev_after_chain; the windows must cover every head exactly once; last_window != 0 on exactly one of them, the one whose
 * launches are ordered behind the CHAIN launches of all others: it advances the device-resident schedule (opt->state)
 * and adds up the loss scalars. ev_after_chain: NULL, or a hipEvent_t this call records between its two launches (what
 * the next window's stream waits for). x_next != NULL (one window of the step only): the next batch is drawn and its
 * features written by guest workgroups of this window's first launch, as nsvd_operator_backward_evd_step_next does.
 * Fused MFMA path, windows whose weight-gradient contraction needs no batch slices
 * (nsvd_backward_head_window_ok), direct or reduced moments: NSVD_EUNSUPPORTED otherwise. Results are bit-identical to
 * the one-call step. */
int nsvd_operator_backward_evd_step_window(const nsvd_model_desc* desc, const nsvd_params* params,
                                           const nsvd_problem* prob, const float* x, int B, const float* f,
                                           const float* Tf, int mask_kind, const float* v, const float* M,
                                           float* moments, int moments_reduced, const void* evd_scratch, int L_total,
                                           int l_offset, float grad_scale, float* loss, const nsvd_params* grads,
                                           const nsvd_rmsprop* opt, void* ws, size_t ws_bytes, int path, int l_begin,
                                           int l_count, int last_window, void* ev_after_chain,
                                           unsigned long long next_seed, unsigned long long next_offset, float* x_next,
                                           void* ws_next, size_t ws_next_bytes, void* stream);

/* nsvd_operator_backward_evd_step that ALSO draws the next batch and writes its Fourier features - what
 * nsvd_operator_sample_features(next_seed, next_offset, x_next, ws_next) does as a launch of its own - as guest
 * workgroups of the backward's first kernel: sampling and features (main_pde.py:92-93, examples/utils.py:139-140)
 * depend on no weight, and that kernel's own workgroups are latency-bound, so the feature launch of the next step and
 * its kernel boundary disappear. ws_next: a second workspace of nsvd_workspace_bytes(desc, B) bytes, distinct from ws;
 * the next nsvd_operator_forward is then called on (x_next, ws_next) with NSVD_FEATURES_READY. MFMA path only
 * (NSVD_EUNSUPPORTED otherwise: call the two entry points separately). Results are bit-identical to the separate calls. */
int nsvd_operator_backward_evd_step_next(const nsvd_model_desc* desc, const nsvd_params* params,
                                         const nsvd_problem* prob, const float* x, int B, const float* f,
                                         const float* Tf, int mask_kind, const float* v, const float* M,
                                         float* moments, int moments_reduced, const void* evd_scratch, int L_total,
                                         int l_offset, float grad_scale, float* loss, const nsvd_params* grads,
                                         const nsvd_rmsprop* opt, void* ws, size_t ws_bytes, int path,
                                         unsigned long long next_seed, unsigned long long next_offset, float* x_next,
                                         void* ws_next, size_t ws_next_bytes, void* stream);

/* The kernel-operator training step's backward half in one call (NestedLoRA.compute_loss_kernel with
 * split_batch = False, methods/nestedlora.py:230-252, followed by loss.backward(); optimizer.step(); ema.update()):
 * after nsvd_model_forward(save_for_backward = 1) produced f = model(x) and the caller computed Tf = Kf from it
 * (nsvd_kernel_apply; no gradient flows through Kf: NestedLoRALossFunctionEVD.backward, methods/nestedlora.py:98-111),
 * this evaluates d loss / d f per sample inside the backward kernels from the moments (as
 * nsvd_operator_backward_evd does: same `moments` / `moments_reduced` / `evd_scratch` conventions), runs autograd's
 * backward of the plain model evaluation and - when opt != NULL - takes the RMSprop (+ EMA) step in the epilogue of
 * the weight-gradient kernel (grads may then be NULL). L_total / l_offset as in nsvd_operator_backward_evd: desc
 * describes the heads [l_offset, l_offset + desc->L) of a model of L_total heads whose f, Tf (B, L_total), moments and
 * masks are global (heads sharded over processes); 0, 0 = all heads. Input dimension up to 64; MFMA path only
 * (128-wide hidden layers, B a multiple of 32): NSVD_EUNSUPPORTED otherwise. ws: the workspace of the
 * nsvd_model_forward call. */
int nsvd_model_backward_evd_step(const nsvd_model_desc* desc, const nsvd_params* params, const float* x, int B,
                                 const float* f, const float* Tf, int mask_kind, const float* v, const float* M,
                                 float* moments, int moments_reduced, const void* evd_scratch, int L_total,
                                 int l_offset, float grad_scale, float* loss, const nsvd_params* grads,
                                 const nsvd_rmsprop* opt, void* ws, size_t ws_bytes, void* stream);

/* One Sketchy-style CDK training step in ONE call: the loop body of examples/cdk/sketchy/main_sketchy.py:180-212 as
 * scripts/exps/sketchy.sh configures it (sgd, momentum 0.9, --clip_grad_norm), AMP branches off:
 *     optimizer.zero_grad(); _, fx, _, fy = method(x, y); loss, *_ = method.compute_loss(fx, fy); loss.backward()
 *     nn.utils.clip_grad_norm_(model.parameters(), max_norm); optimizer.step()
 * with model = HeteroNetwork(two get_mlp towers, Identity projectors, mu, regularize_mode) (examples/models/siam.py:132-166)
 * and method = NestedLoRAForCDK(model, neigs = d2, ...) (methods/nestedlora.py:335-378). Stages: nsvd_tower_forward x 2
 * (running statistics updated), nsvd_row_normalize_forward x 2 (radius sqrt(mu)), nsvd_cdk_loss_forward / _backward,
 * nsvd_row_normalize_backward x 2, nsvd_tower_backward x 2, then the total gradient norm over all 16 parameter tensors
 * in a fixed summation order, torch's clip coefficient min(1, max_grad_norm / (norm + 1e-6)) (max_grad_norm <= 0: no
 * clipping) and torch.optim.SGD's momentum update (no dampening, nesterov or weight decay):
 *     g <- coef g;  buf <- g (first_step) or momentum buf + g;  p <- p - lr buf
 * towers[2] (x tower, y tower): parameters and running statistics, UPDATED IN PLACE; momentum_bufs[2]: the optimiser's
 * momentum buffers in the same layouts (rm / rv fields unused). lr is the already scheduled value. x, y: (B, d0).
 * v (d2 + first), M ((d2 + first)^2): nesting masks. loss[0..3] = {loss, operator term, metric term, total gradient
 * norm before clipping}. rs_joint (B) / rs_indep (B (B - 1)): the loss's diagnostics, or NULL. Shapes as for
 * nsvd_tower_forward. ws: nsvd_cdk_step_workspace_bytes (0 for an unsupported description). */
typedef struct nsvd_cdk_step_desc {
    int32_t B, d0, d1, d2;
    float slope, bn_eps, bn_momentum;
    float mu;
    int32_t normalize_mode;       /* NSVD_NORMALIZE_* */
    int32_t set_first_mode_const;
    double lr, momentum, max_grad_norm;
    int32_t first_step;
    int32_t gemm_bf16;            /* != 0: the towers in mixed precision (nsvd_tower_forward; bit 1: this workspace's
                                   * bfloat16 weight copies are those the previous nsvd_cdk_step call left; bit 4,
                                   * value 16: the half type is IEEE float16 instead of bfloat16) */
    int32_t reserved0;
    void* grad_scaler;            /* NULL, or a DEVICE nsvd_grad_scaler: the step runs with loss scaling (below) */
} nsvd_cdk_step_desc;
/* torch.cuda.amp.GradScaler's state and arithmetic on the device - the reference's AMP branch, on by default in the
 * Sketchy script (examples/cdk/sketchy/main_sketchy.py:161 scaler = GradScaler(enabled=use_amp); :194-208
 * scaler.scale(loss).backward(); scaler.unscale_(optimizer); clip_grad_norm_; scaler.step(optimizer); scaler.update();
 * lr_scheduler.step() on EVERY iteration, skipped or not, :205-206 - this script has no scheduler gate, unlike the PDE
 * loop's examples/operator/__init__.py:66-72: desc.lr stays the caller's scheduled value of the iteration). With
 * desc.grad_scaler set, nsvd_cdk_step
 *   - multiplies the loss gradient by `scale` where the backward starts (every stored 16-bit gradient is scaled),
 *   - takes found_inf = the gradient norm (of the scaled gradients, float32) is not finite,
 *   - found_inf: NO parameter, momentum buffer or weight copy is written (scaler.step skips optimizer.step()), scale *=
 *     backoff_factor, growth_tracker = 0, steps_skipped += 1;
 *   - otherwise: gradients * (1 / scale), then the clip coefficient and the SGD update as without scaling; steps_ok += 1,
 *     growth_tracker += 1 and at growth_interval: scale *= growth_factor, growth_tracker = 0 (GradScaler.update()).
 *   desc.first_step is ignored: the momentum buffers start with the first step TAKEN (steps_ok == 0), as torch.optim.SGD
 *   creates them in its first executed step().
 * The loss values reported are unscaled. Needs the mixed-precision step with the fused narrow end (B % 8 == 0,
 * d2 % 64 == 0, d2 <= 1024), NSVD_EUNSUPPORTED otherwise. Read the state back with a device-to-host copy. */
typedef struct nsvd_grad_scaler {
    float scale;
    float growth_factor;      /* torch default 2 */
    float backoff_factor;     /* 0.5 */
    int32_t growth_interval;  /* 2000 */
    int32_t growth_tracker;
    int32_t steps_ok;
    int32_t steps_skipped;
    int32_t last_found_inf;
} nsvd_grad_scaler;
int nsvd_grad_scaler_init(void* state, float init_scale, float growth_factor, float backoff_factor, int growth_interval,
                          void* stream);
size_t nsvd_cdk_step_workspace_bytes(const nsvd_cdk_step_desc* desc);
int nsvd_cdk_step(const nsvd_cdk_step_desc* desc, const float* x, const float* y, const nsvd_tower_params* towers,
                  const nsvd_tower_params* momentum_bufs, const float* v, const float* M, float* loss,
                  float* rs_joint, float* rs_indep, void* ws, size_t ws_bytes, void* stream);

/* C = A B^T on the bf16 MFMA, float32 accumulation: the contraction of the mixed-precision towers (csrc/gemm16.h) as an
 * entry point of its own - what torch.matmul on bfloat16 tensors is to the reference's autocast branch
 * (examples/cdk/sketchy/main_sketchy.py:182; the reference's half type is float16). A, B hold bfloat16 values.
 *   a_kstrided == 0: A is (M, lda), row m = the K contraction values of output row m (contiguous)
 *   a_kstrided != 0: A is (K, lda), row k = the M values of contraction index k (A^T as stored: no transposed copy)
 *   likewise B / b_kstrided with N columns of C. (A strided with B contiguous is not built.)
 * C (M, ldc): float32, or bfloat16 when out_bf16; bias (N) is added per column when not NULL. slices > 1: split-K,
 * slice s contracts k in [s K / slices, (s + 1) K / slices) into C + s * slice_stride (elements; bias goes to every
 * slice: pass NULL). sumsq: NULL, or (M / 256) (N / 128) slices floats - per tile, the sum of squares of the
 * float32 values BEFORE any bfloat16 rounding of the store (the gradient norm is taken of the float32 gradient).
 * M % 256 == 0, N % 128 == 0, K % (64 slices) == 0, lda, ldb % 8 == 0, ldc % 4 == 0, 16-byte aligned pointers. */
int nsvd_gemm_bf16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb,
                   long ldc, int a_kstrided, int b_kstrided, int out_bf16, int slices, long slice_stride, float* sumsq,
                   void* stream);
/* out[i] = bfloat16(in[i]) (round to nearest even), n % 8 == 0, 16-byte aligned pointers */
int nsvd_to_bf16(const float* in, void* out, size_t n, void* stream);

/* Measurement aid (bench.py): record the two hipEvent_t handles immediately before / after the
 * DOMINANT kernel of the next nsvd_operator_forward call made by this host thread (the fused MFMA
 * forward kernel, or the layer-0 GEMM on the generic path), on that call's stream - or, whichever
 * comes first, around the first contraction (X W1^T) of the next nsvd_tower_forward / the gathered-row
 * contraction of the next nsvd_kernel_apply. One-shot; pass NULLs to clear. Per-thread state; the
 * compute entry points themselves stay stateless. */
int nsvd_profile_next_forward(void* ev_start, void* ev_stop);

#ifdef __cplusplus
}
#endif
#endif /* NSVD_H */
