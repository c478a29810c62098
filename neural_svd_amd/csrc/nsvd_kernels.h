// Internal (C++) launch interface between the translation units of libnsvd_hip.so.
#pragma once
#include "nsvd_common.h"

// ---- measurement hook (operator_api.hip): bracket the dominant forward kernel with events ----------
void nsvd_prof_begin(hipStream_t s);
void nsvd_prof_end(hipStream_t s);

// Fused-path features (fourier.hip): phi (B, 2m) sample-major features of the CENTRE rows from one double-accurate
// sincos per (sample, frequency); phiTc (2m, B) feature-major copy, or null; sctab (D, 2, m) = cos / sin of
// eps * fourier_B, from which the forward kernel builds the even / odd perturbation rows of the shifted stencil points.
// sampler != null: the coordinates are drawn inside the kernel (N(0, sigma^2), counter-based) and stored to xout.
int nsvd_fourier_stencil(const float* x, const float* fourier_B, float* phi, float* phiTc, float* sctab, int B, int D,
                         int m, float eps, const NsvdSampler* sampler, float* xout, hipStream_t s);
// generic path, stencil mode: feature-major phiT (2 m, ldr) with the 2 D shifted row blocks in even / odd form
int nsvd_fourier_features_evenodd(const float* x, const float* fourier_B, float* phiT, int B, int D, int m, float eps,
                                  int ldr, hipStream_t s);
// centre features only, any input dimension D (plain model evaluation): phi (B, 2m), phiTc (2m, B) or null
int nsvd_fourier_plain(const float* x, const float* fourier_B, float* phi, float* phiTc, int B, int D, int m,
                       hipStream_t s);
// stand-alone draw of the same values (generic path)
int nsvd_sample_launch(const NsvdSampler& smp, float* x, int B, int D, hipStream_t s);

// ---- generic strided batched GEMM (gemm_generic.hip) -------------------------------------------
//   C[g][i][j] = epi( sum_k A[g][i*sAm + k*sAk] * B[g][k*sBk + j*sBn] + bias[g][i] )
// epi: optional multiply by sigmoid(z) GIVEN a = softplus(z) stored at Z[g][i*sZm + j] (nsvd_sigmoid_from_softplus).
struct NsvdGemm {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0, batch = 1;
    long sAm = 0, sAk = 0, sBk = 0, sBn = 0, sCm = 0;
    long bA = 0, bB = 0, bC = 0;
    const float* bias = nullptr;
    long bBias = 0;
    const float* Z = nullptr;
    long sZm = 0, bZ = 0;
    int sigmoid_mul = 0;
    // optional: rowsum[g * bRowsum + i] = sum_k A[g][i][k] (the bias gradient beside a weight gradient); computed by the
    // launch itself where it can (nsvd_gemm_generic's rowsum_done), by nsvd_rowsum otherwise
    float* rowsum = nullptr;
    long bRowsum = 0;
    // stencil columns in even / odd form (eo_cols = samples per stencil block, 0: off): column j belongs to block
    // e = j / eo_cols (0 the centre, 1 + 2 d / 2 + 2 d the even / odd perturbation along d). The bias joins the centre
    // block only.
    int eo_cols = 0;
};
int nsvd_gemm_generic(const NsvdGemm& g, hipStream_t s, bool* rowsum_done = nullptr);
// in place, z (rows x nst * B, stencil blocks of B columns): pre-activations -> activations; block 0 softplus, blocks
// 1 + 2 d / 2 + 2 d the even / odd parts of the softplus of the shifted pair (nst = 1: plain softplus)
int nsvd_softplus_inplace(float* z, long rows, int B, int nst, hipStream_t s);

// out[g][i] = sum_j in[g][i*ld + j], j < n   (bias gradients)
int nsvd_rowsum(const float* in, float* out, int rows, int n, long ld, hipStream_t s);

// ---- FD Hamiltonian epilogue + its backward head (fd_epilogue.hip) ------------------------------
// base: (L, ldr) head outputs at the E*B stencil rows -> f, Tf (B, L); optionally jac, dsc (B, L):
//   jac = d f / d base(centre), dsc = d f / d scales_l per row (ExponentialMask only).
// evenodd != 0: rows 1 + 2 d / 2 + 2 d of `base` hold the EVEN / ODD perturbations of the head output along direction d
// (base(x +- eps e_d) = base[0] + even_d +- odd_d: the fused kernels' split-stencil form) instead of the point values
int nsvd_fd_epilogue(const float* base, int ldr, const float* x, const float* scales, const nsvd_problem& prob,
                     int B, int D, int L, float* f, float* Tf, float* jac, float* dsc, hipStream_t s, int evenodd = 0);
// dzT[l][b] = df[b][l] * jac[b][l]; dscales[l] = sum_b df[b][l] * dsc[b][l] (when dscales != null)
int nsvd_head_backward(const float* df, const float* jac, const float* dsc, int B, int L, float* dzT,
                       float* dscales, hipStream_t s);

// out[b][l] = c * base[l*ldr + b] * exp(-|x_b| / scales[l])   (WaveFunctions.forward at the centre rows)
// optional jac = d out / d base, dsc = d out / d scales (both (B, L)) for nsvd_model_backward
int nsvd_model_out(const float* base, int ldr, const float* x, const float* scales, float c, int B, int D, int L,
                   float* out, float* jac, float* dsc, hipStream_t s);

// reduce the per-chunk partial moments of nsvd_evd_partial into the (2 L^2 + 1) vector (evd_loss.hip)
int nsvd_evd_reduce_partials(const void* scratch, int B, int L, float* moments, hipStream_t s);

// ---- fused MFMA path (pmlp_fwd.hip, pmlp_bwd.hip) -----------------------------------------------------------
// plain model evaluation out = c * model(x) on the fused kernels (E = 1 instance; input dimension up to 64);
// save != 0 keeps what nsvd_fused_backward needs (dout plays the role of df)
bool nsvd_fused_model_supported(const nsvd_model_desc& d, int B);
int nsvd_fused_model_forward(const nsvd_model_desc& d, const nsvd_params& p, const float* x, int B, float c,
                             float* out, void* ws, int save, hipStream_t s);
bool nsvd_fused_supported(const nsvd_model_desc& d, int B, bool exact = false);
size_t nsvd_fused_workspace_bytes(const nsvd_model_desc& d, int B);
// Fourier features of x into the fused path's workspace (phi, and phiT_c when save != 0)
int nsvd_fused_features(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                        int B, void* ws, int save, hipStream_t s, const NsvdSampler* sampler = nullptr,
                        float* xout = nullptr);
// save: bit 0 = keep what the backward needs, bit 1 = features already prepared by nsvd_fused_features
int nsvd_fused_forward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                       int B, float* f, float* Tf, void* ws, int save, hipStream_t s, int bf3 = 0);
// batch slices of the weight-gradient kernel (1: every tile contracts the whole batch)
int nsvd_fused_wgrad_slices(const nsvd_model_desc& d, int B);
int nsvd_fused_stream_bwd_slices(const nsvd_model_desc& d, int B);  // > 0: the streaming backward may take the step
int nsvd_fused_backward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                        int B, const float* df, const nsvd_params& g, void* ws, hipStream_t s);
struct NsvdEvdIn;  // evd_math.h
// backward with d loss / d f derived from the EVD moments inside the chain kernel (no df round trip)
// g: where to store the gradients (null: not stored); opt: RMSprop + EMA applied in the weight-gradient kernel's
// epilogue, parameters updated in place (null: no step). At least one of the two.
struct NsvdOptStep;  // opt_math.h
struct NsvdHyper;
// [l_begin, l_begin + l_count): the heads whose gradients this call produces (l_count = 0: all of them)
// next: draw the NEXT batch (sampler, written to next->x) and write its features into the workspace next->ws as guest
// workgroups of the chain kernel (what nsvd_fused_features(sampler) does as a launch of its own), or null
struct NsvdNextBatch {
    NsvdSampler smp;
    float* x;     // (B, D) receives the drawn coordinates
    void* ws;     // the other workspace set (same size as the step's)
    float eps;    // the problem's finite-difference eps (<= 0: exact-Laplacian constants)
};
// window_of_step: this call is one of several head windows of ONE fused step; not_last: it neither advances the device-
// resident schedule nor adds up the step's loss; ev_after_chain (hipEvent_t or null): recorded between its two launches
int nsvd_fused_backward_evd(const nsvd_model_desc& d, const nsvd_params& p, int B, const NsvdEvdIn& evd,
                            const nsvd_params* g, const NsvdOptStep* opt, void* ws, hipStream_t s, int l_begin = 0,
                            int l_count = 0, const NsvdNextBatch* next = nullptr, int window_of_step = 0,
                            int not_last = 0, void* ev_after_chain = nullptr);
bool nsvd_fused_backward_window_ok(const nsvd_model_desc& d, int B, int l_count);
// stand-alone optimiser launch over n contiguous floats (optimizer.hip)
int nsvd_rmsprop_launch(float* p, const float* grad, float* sq, float* ema, size_t n, const NsvdHyper& h,
                        hipStream_t s, nsvd_step_state* state = nullptr, int advance = 0);

// the same update over a table of tensors in ONE launch (start[] is filled by the launcher; every pointer 16-byte aligned,
// else NSVD_EUNSUPPORTED and nothing is launched)
constexpr int NSVD_OPT_TABLE_MAX = 2 * NSVD_MAX_LAYERS + 1;
struct NsvdOptTable {
    float* p[NSVD_OPT_TABLE_MAX];
    const float* g[NSVD_OPT_TABLE_MAX];
    float* sq[NSVD_OPT_TABLE_MAX];
    float* ema[NSVD_OPT_TABLE_MAX];  // all null: no EMA
    size_t n[NSVD_OPT_TABLE_MAX];
    size_t start[NSVD_OPT_TABLE_MAX + 1];  // prefix sums in float4 groups
    int count;
};
int nsvd_rmsprop_table_launch(NsvdOptTable& t, const NsvdHyper& h, hipStream_t s);

// ---- CDK towers (tower.hip): the backward with per-workgroup sums of squares of the two weight-gradient contractions
// (cdk_step.hip clips the global gradient norm without another pass over them)
int nsvd_tower_sumsq_count(int d0, int d1, int d2, int gemm_bf16 = 0);
int nsvd_tower_backward_sumsq(const float* x, const nsvd_tower_params* p, const float* dz, int B, int d0, int d1,
                              int d2, float slope, int gemm_bf16, const nsvd_tower_params* grads, void* ws,
                              size_t ws_bytes, float* sumsq, void* stream);
// mixed precision: both towers of a step through every launch together (tower.hip); sumsq[t]: that tower's partials
// flags: the gemm_bf16 bits, plus NSVD_TOWER16_WIDE_ONLY: stop behind the split-K partial products of Y2 (forward; z
// unused) / start behind dY2 (backward: the caller has written the bfloat16 dY2 into the workspace, dz unused) - the
// narrow end is then the caller's (cdk_narrow.hip)
constexpr int NSVD_TOWER16_WIDE_ONLY = 4;
// backward: sumsq[t] continues behind the contractions' partials with one float per 64-column strip of the first
// BatchNorm's backward (d1 / 64): the squares of the b1 / g1 / be1 gradients that strip wrote
constexpr int NSVD_TOWER16_SMALL_SUMSQ = 8;
// the half type is IEEE float16 instead of bfloat16 (a gemm_bf16 bit of include/nsvd.h: forward and backward alike)
constexpr int NSVD_TOWER16_F16 = 16;
int nsvd_tower16_forward_pair(const float* const* x, const nsvd_tower_params* const* p, int B, int d0, int d1, int d2,
                              float slope, float eps, float momentum, int update_running, int flags, float* const* z,
                              void* const* ws, size_t ws_bytes, hipStream_t s);
int nsvd_tower16_backward_pair(const float* const* x, const nsvd_tower_params* const* p, const float* const* dz, int B,
                               int d0, int d1, int d2, float slope, const nsvd_tower_params* const* grads,
                               void* const* ws, size_t ws_bytes, float* const* sumsq, hipStream_t s, int flags = 0);
void nsvd_tower16_weight_copies(int B, int d0, int d1, int d2, void* ws, void** W1h, void** W2h);

// ---- the narrow end of a CDK step for both towers at once (cdk_narrow.hip): split-K sum + bias, BatchNorm2 (training),
// normalize; and the backward of the three. Per tower t < nt.
struct NsvdNarrowFwd {
    int nt, B, N, S;
    size_t slice_stride;         // floats between the split-K slices of Y2p
    const float* Y2p[2];         // (S, B, N) partial products (no bias)
    const float* bias[2];        // b2 (N) or null
    const float* gamma[2];
    const float* beta[2];
    float* running_mean[2];      // updated in place, or null
    float* running_var[2];
    float* mean[2];              // (N) saved for the backward
    float* invstd[2];
    float* Y2[2];                // (B, N) the summed pre-normalisation output (kept for the backward)
    float* z[2];                 // (B, N) BatchNorm output
    float* e[2];                 // (B, N) normalize(z)
    float* part;                 // scratch: nsvd_narrow_scratch_floats
    float eps, momentum, r_up;
    int sphere;                  // 0: l2_ball, 1: l2_sphere
};
struct NsvdNarrowBwd {
    int nt, B, N;
    const float* z[2];           // (B, N) the forward's BatchNorm output
    const float* ge[2];          // (B, N) gradient w.r.t. normalize(z)
    const float* Y2[2];
    const float* mean[2];
    const float* invstd[2];
    const float* gamma[2];
    float* dz[2];                // (B, N) scratch: gradient w.r.t. z
    void* dY[2];                 // (B, N) gradient w.r.t. Y2: float32 (dy_bf16 0), bfloat16 (1) or float16 (2)
    int dy_bf16;
    const float* loss_scale;     // null, or a DEVICE scalar the incoming gradient ge is multiplied by as it is read (the
                                 // GradScaler's loss scale: cdk_step.hip) - everything downstream is then scaled
    float* dgamma[2];
    float* dbeta[2];
    float* dbias[2];             // gradient of b2
    float* sumsq[2];             // null, or nsvd_narrow_sumsq_count(N) floats per tower: squares of the b2 / g2 / be2
                                 // gradients, per group of columns
    float* part;                 // scratch: nsvd_narrow_scratch_floats
    float* m1[2];                // (set by nsvd_narrow_backward: inside `part`)
    float* m2[2];
    float r_up;
    int sphere;
};
// the CDK loss forward without its one-block reduction launch: the per-block partials of the two loss terms, for a later
// kernel of the same stream to add (nsvd_cdk_loss_sum: the reduction's own order) - cdk_step.hip's optimiser kernel
struct NsvdCdkLossParts {
    const float *part_op, *part_met;
    int nstage, nfin, B;
};
int nsvd_cdk_loss_forward_parts(const float* f, const float* g, const float* batch_weights, const float* v,
                                const float* M, int B, int L, int set_first_mode_const, float* rs_joint,
                                float* rs_indep, void* ws, size_t ws_bytes, NsvdCdkLossParts* parts, hipStream_t s);
// loss[0..2] = {loss, operator term, metric term} from the partials, added in index order by one workgroup of 256
// threads (red: 8 floats of LDS - red[0..7] are written); every thread of the workgroup must call it
__device__ __forceinline__ void nsvd_cdk_loss_sum(const NsvdCdkLossParts& lp, float* red, float* __restrict__ loss) {
    float so = 0.f, sm = 0.f;
    for (int i = threadIdx.x; i < lp.nstage; i += 256) so += lp.part_op[i];
    for (int i = threadIdx.x; i < lp.nfin; i += 256) sm += lp.part_met[i];
    so = nsvd_wave_sum(so);
    sm = nsvd_wave_sum(sm);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = so;
        red[4 + (threadIdx.x >> 6)] = sm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float lop = -2.0f * (red[0] + red[1] + red[2] + red[3]) / (float)lp.B;
        const float met = red[4] + red[5] + red[6] + red[7];
        loss[0] = lop + met;
        loss[1] = lop;
        loss[2] = met;
    }
}
size_t nsvd_narrow_scratch_floats(int nt, int B, int N);
int nsvd_narrow_sumsq_count(int N);
bool nsvd_narrow_supported(int nt, int B, int N);
int nsvd_narrow_forward(const NsvdNarrowFwd& a, hipStream_t s);
int nsvd_narrow_backward(const NsvdNarrowBwd& a, hipStream_t s);
// a mixed-precision tower's narrow-end buffers inside its workspace, and the split-K slice count of a launch of nt towers
struct NsvdTowerNarrowViews {
    float *Y2p, *Y2, *mean2, *inv2;
    void* dY2h;
    int S;
    size_t slice_stride;
};
NsvdTowerNarrowViews nsvd_tower16_narrow_views(int nt, int B, int d0, int d1, int d2, void* ws);
