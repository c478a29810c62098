// extern "C" entry points for the operator forward / backward: argument validation, workspace
// carving, and the choice between the fused MFMA kernels (pmlp_fwd.hip, pmlp_bwd.hip) and the generic
// layer-by-layer path implemented here on top of gemm_generic.hip / fd_epilogue.hip.
#include <stdlib.h>
#include <string.h>
#include "nsvd_kernels.h"
#include "evd_math.h"
#include "opt_math.h"

static thread_local hipEvent_t g_prof_start = nullptr;
static thread_local hipEvent_t g_prof_stop = nullptr;

void nsvd_prof_begin(hipStream_t s) {
    if (g_prof_start) (void)hipEventRecord(g_prof_start, s);
    g_prof_start = nullptr;
}
void nsvd_prof_end(hipStream_t s) {
    if (g_prof_stop) (void)hipEventRecord(g_prof_stop, s);
    g_prof_stop = nullptr;
}

extern "C" int nsvd_profile_next_forward(void* ev_start, void* ev_stop) {
    g_prof_start = (hipEvent_t)ev_start;
    g_prof_stop = (hipEvent_t)ev_stop;
    return 0;
}

namespace {

struct GenericWs {
    float* phiT;                      // (F, R)
    float* z[NSVD_MAX_LAYERS];        // z[i]: (L, h_i, R) ACTIVATIONS a_i for i < nl-1 ; z[nl-1] = the model output (L, R)
    float* jac;                       // (B, L)
    float* dsc;                       // (B, L)
    float* dz[2];                     // ping-pong (L, hmax, B)
    float* df;                        // (B, L) for nsvd_operator_backward_evd on the generic path
    int R;                            // rows per (head, unit): erows * B, erows = stencil blocks laid out
    size_t bytes;
};

// plain model evaluation (nsvd_model_forward / _backward) has no stencil: any input dimension up to 64
constexpr int MODEL_MAX_D = 64;
int validate(const nsvd_model_desc* d, int max_d = 4) {
    if (!d) return NSVD_EINVAL;
    if (d->L <= 0 || d->D <= 0 || d->m <= 0) return NSVD_EINVAL;
    if (d->nlayers < 1 || d->nlayers > NSVD_MAX_LAYERS) return NSVD_EINVAL;
    for (int i = 0; i < d->nlayers; ++i)
        if (d->dims[i] <= 0) return NSVD_EINVAL;
    if (d->dims[d->nlayers - 1] != 1) return NSVD_EINVAL;
    if (d->D > max_d) return NSVD_EUNSUPPORTED;
    return 0;
}

// erows: stencil blocks the layout holds (1 + 2D for the operator, 1 for plain model evaluation)
GenericWs carve(const nsvd_model_desc& d, int B, void* base, int erows = 0) {
    GenericWs w;
    memset(&w, 0, sizeof(w));
    const size_t E = erows > 0 ? (size_t)erows : 1 + 2 * (size_t)d.D, R = E * B, F = 2 * (size_t)d.m;
    w.R = (int)R;
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float* q = (float*)(p + off);
        off += nsvd_align(nfloats * sizeof(float));
        return q;
    };
    w.phiT = take(F * R);
    int hmax = 1;
    for (int i = 0; i < d.nlayers; ++i) {
        w.z[i] = take((size_t)d.L * d.dims[i] * R);
        if (d.dims[i] > hmax) hmax = d.dims[i];
    }
    w.jac = take((size_t)B * d.L);
    w.dsc = take((size_t)B * d.L);
    w.dz[0] = take((size_t)d.L * hmax * B);
    w.dz[1] = take((size_t)d.L * hmax * B);
    w.df = take((size_t)B * d.L);
    w.bytes = off;
    return w;
}

// exact: the exact-Laplacian mode (prob->eps <= 0), whose D + 2 jet streams fit the MFMA path up to D = 3
bool want_fused(const nsvd_model_desc& d, int B, int path, bool exact = false) {
    if (path == NSVD_PATH_GENERIC) return false;  // NSVD_PATH_FUSED and NSVD_PATH_FUSED_BF16X3 both need the MFMA path
    return nsvd_fused_supported(d, B, exact);
}

int check_params(const nsvd_model_desc& d, const nsvd_params* p, bool need_fourier) {
    if (!p) return NSVD_EINVAL;
    if (need_fourier && !p->fourier_B) return NSVD_EINVAL;
    for (int i = 0; i < d.nlayers; ++i)
        if (!p->W[i] || !p->b[i]) return NSVD_EINVAL;
    if (d.has_exp_mask && !p->scales) return NSVD_EINVAL;
    return 0;
}

// Fourier map + every ParallelMLP layer for the first `nst` stencil blocks (nst = E: all rows, 1: centre only)
int generic_mlp(const nsvd_model_desc& d, const nsvd_params& p, const float* x, int B, float eps, int nst,
                const GenericWs& w, hipStream_t s, bool features_ready = false) {
    const int R = w.R, F = 2 * d.m;
    int rc = 0;
    // all stencil rows: the shifted row blocks in EVEN / ODD form - features, pre-activations and head
    // outputs of block 1 + 2 d / 2 + 2 d are the even / odd perturbations along d; centre rows only (nst = 1): plain
    const bool eo = nst > 1;
    if (!features_ready)
        rc = eo ? nsvd_fourier_features_evenodd(x, p.fourier_B, w.phiT, B, d.D, d.m, eps, R, s)
                : nsvd_fourier_features(x, p.fourier_B, w.phiT, B, d.D, d.m, eps, nst, R, s);
    if (rc) return rc;
    int kin = F;
    for (int i = 0; i < d.nlayers; ++i) {
        NsvdGemm g;
        g.batch = d.L;
        g.M = d.dims[i]; g.N = nst * B; g.K = kin;
        g.A = p.W[i]; g.sAm = kin; g.sAk = 1; g.bA = (long)d.dims[i] * kin;
        g.B = (i == 0) ? w.phiT : w.z[i - 1]; g.sBk = R; g.sBn = 1; g.bB = (i == 0) ? 0 : (long)kin * R;
        g.C = w.z[i]; g.sCm = R; g.bC = (long)d.dims[i] * R;
        g.bias = p.b[i]; g.bBias = d.dims[i];
        g.eo_cols = eo ? B : 0;
        rc = nsvd_gemm_generic(g, s);
        if (rc) return rc;
        // hidden layers: pre-activations -> activations in place (what the next layer, the weight gradient and the data
        // gradient's sigmoid factor read); the last layer's output stays as it is
        if (i + 1 < d.nlayers) {
            rc = nsvd_softplus_inplace(w.z[i], (long)d.L * d.dims[i], B, nst, s);
            if (rc) return rc;
        }
        kin = d.dims[i];
    }
    return 0;
}

int generic_forward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x, int B,
                    float* f, float* Tf, void* ws, hipStream_t s, bool features_ready = false) {
    const GenericWs w = carve(d, B, ws);
    const int E = 1 + 2 * d.D, R = E * B;
    // (bench.py's event bracket: the WHOLE forward - features, every layer, the finite-difference epilogue - as the fused
    // forward kernel is one launch doing all of it)
    nsvd_prof_begin(s);
    int rc = generic_mlp(d, p, x, B, prob.eps, E, w, s, features_ready);
    if (rc) return rc;
    rc = nsvd_fd_epilogue(w.z[d.nlayers - 1], R, x, d.has_exp_mask ? p.scales : nullptr, prob, B, d.D, d.L, f, Tf,
                          w.jac, w.dsc, s, 1);
    nsvd_prof_end(s);
    return rc;
}

int generic_backward(const nsvd_model_desc& d, const nsvd_params& p, const float* x, int B, const float* df,
                     const nsvd_params& g, void* ws, hipStream_t s, int erows = 0) {
    (void)x;
    const GenericWs w = carve(d, B, ws, erows);
    const int R = w.R, F = 2 * d.m;
    int cur = 0;
    int rc = nsvd_head_backward(df, w.jac, w.dsc, B, d.L, w.dz[cur], d.has_exp_mask ? g.scales : nullptr, s);
    if (rc) return rc;
    for (int i = d.nlayers - 1; i >= 0; --i) {
        const int hi = d.dims[i];
        const int kin = (i == 0) ? F : d.dims[i - 1];
        // weight gradient: dW_i[l][n][k] = sum_b dz_i[l][n][b] * a_{i-1}[l][k][b]   (w.z[i - 1] holds a_{i-1})
        NsvdGemm wg;
        wg.batch = d.L;
        wg.M = hi; wg.N = kin; wg.K = B;
        wg.A = w.dz[cur]; wg.sAm = B; wg.sAk = 1; wg.bA = (long)hi * B;
        wg.B = (i == 0) ? w.phiT : w.z[i - 1]; wg.sBk = 1; wg.sBn = R; wg.bB = (i == 0) ? 0 : (long)kin * R;
        wg.C = g.W[i]; wg.sCm = kin; wg.bC = (long)hi * kin;
        // bias gradient = row sums of dz_i: inside the weight-gradient launch where the kernel can, its own launch otherwise
        wg.rowsum = g.b[i]; wg.bRowsum = hi;
        bool rowsum_done = false;
        rc = nsvd_gemm_generic(wg, s, &rowsum_done);
        if (rc) return rc;
        if (!rowsum_done) {
            rc = nsvd_rowsum(w.dz[cur], g.b[i], d.L * hi, B, B, s);
            if (rc) return rc;
        }
        if (i > 0) {
            // data gradient: dz_{i-1}[l][k][b] = (sum_n W_i[l][n][k] dz_i[l][n][b]) * sigmoid(z_{i-1}[l][k][b]), the sigmoid
            // from the stored activation (1 - e^-a)
            NsvdGemm dg;
            dg.batch = d.L;
            dg.M = kin; dg.N = B; dg.K = hi;
            dg.A = p.W[i]; dg.sAm = 1; dg.sAk = kin; dg.bA = (long)hi * kin;
            dg.B = w.dz[cur]; dg.sBk = B; dg.sBn = 1; dg.bB = (long)hi * B;
            dg.C = w.dz[cur ^ 1]; dg.sCm = B; dg.bC = (long)kin * B;
            dg.Z = w.z[i - 1]; dg.sZm = R; dg.bZ = (long)kin * R;
            dg.sigmoid_mul = 1;
            rc = nsvd_gemm_generic(dg, s);
            if (rc) return rc;
            cur ^= 1;
        }
    }
    return 0;
}

}  // namespace

extern "C" int nsvd_abi_version(void) { return NSVD_ABI_VERSION; }

extern "C" const char* nsvd_path_name(const nsvd_model_desc* desc, int B, int path) {
    if (validate(desc) != 0 || B <= 0) return "invalid";
    return want_fused(*desc, B, path) ? "fused_mfma" : "generic";
}

extern "C" int nsvd_step_emits_planes(const nsvd_model_desc* desc, int B, int path) {
    if (validate(desc) != 0 || B <= 0 || path != NSVD_PATH_FUSED_BF16X3) return 0;
    const char* e = getenv("NSVD_PLANES_FROM_STEP");
    if (e && atoi(e) == 0) return 0;
    // (the streaming backward never runs the kernel that writes the planes)
    return nsvd_fused_supported(*desc, B, false) && nsvd_fused_wgrad_slices(*desc, B) == 1 &&
                   nsvd_fused_stream_bwd_slices(*desc, B) == 0 ? 1 : 0;
}

extern "C" const char* nsvd_path_name_for(const nsvd_model_desc* desc, const nsvd_problem* prob, int B, int path) {
    if (validate(desc) != 0 || B <= 0 || !prob) return "invalid";
    const bool exact = !(prob->eps > 0.f);
    if (want_fused(*desc, B, path, exact)) return "fused_mfma";
    return exact ? "unsupported" : "generic";
}

extern "C" size_t nsvd_model_workspace_bytes(const nsvd_model_desc* desc, int B) {
    if (validate(desc, MODEL_MAX_D) != 0 || B <= 0) return 0;
    const size_t gen = carve(*desc, B, nullptr, 1).bytes;
    const size_t fus = nsvd_fused_model_supported(*desc, B) ? nsvd_fused_workspace_bytes(*desc, B) : 0;
    return gen > fus ? gen : fus;
}

extern "C" size_t nsvd_workspace_bytes(const nsvd_model_desc* desc, int B) {
    if (validate(desc) != 0 || B <= 0) return 0;
    const size_t gen = carve(*desc, B, nullptr).bytes;
    const size_t fus = nsvd_fused_supported(*desc, B, true) ? nsvd_fused_workspace_bytes(*desc, B) : 0;
    return gen > fus ? gen : fus;
}

extern "C" int nsvd_operator_forward(const nsvd_model_desc* desc, const nsvd_params* params,
                                     const nsvd_problem* prob, const float* x, int B, float* f, float* Tf, void* ws,
                                     size_t ws_bytes, int save_for_backward, int path, void* stream) {
    int rc = validate(desc);
    if (rc) return rc;
    if (!prob || !x || !f || !Tf || !ws || B <= 0) return NSVD_EINVAL;
    if (prob->potential != NSVD_POT_HYDROGEN && prob->potential != NSVD_POT_HARMONIC) return NSVD_EINVAL;
    rc = check_params(*desc, params, true);
    if (rc) return rc;
    if (ws_bytes < nsvd_workspace_bytes(desc, B)) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    const bool fused = want_fused(*desc, B, path, !(prob->eps > 0.f));
    if ((path == NSVD_PATH_FUSED || path == NSVD_PATH_FUSED_BF16X3) && !fused) return NSVD_EUNSUPPORTED;
    // eps <= 0 selects the exact Laplacian (reference diff_ops.py:7): forward-mode jets, MFMA path only
    if (!(prob->eps > 0.f) && !fused) return NSVD_EUNSUPPORTED;
    const bool ready = (save_for_backward & NSVD_FEATURES_READY) != 0;
    const bool planes = (save_for_backward & NSVD_W_PLANES_READY) != 0 && path == NSVD_PATH_FUSED_BF16X3;
    if (fused)
        return nsvd_fused_forward(*desc, *params, *prob, x, B, f, Tf, ws,
                                  (save_for_backward & 1) | (ready ? 2 : 0) | (planes ? 4 : 0), (hipStream_t)stream,
                                  path == NSVD_PATH_FUSED_BF16X3);
    return generic_forward(*desc, *params, *prob, x, B, f, Tf, ws, (hipStream_t)stream, ready);
}

extern "C" int nsvd_operator_features(const nsvd_model_desc* desc, const nsvd_params* params,
                                      const nsvd_problem* prob, const float* x, int B, void* ws, size_t ws_bytes,
                                      int save_for_backward, int path, void* stream) {
    int rc = validate(desc);
    if (rc) return rc;
    if (!prob || !x || !ws || !params || !params->fourier_B || B <= 0) return NSVD_EINVAL;
    if (ws_bytes < nsvd_workspace_bytes(desc, B)) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    const bool fused = want_fused(*desc, B, path, !(prob->eps > 0.f));
    if ((path == NSVD_PATH_FUSED || path == NSVD_PATH_FUSED_BF16X3) && !fused) return NSVD_EUNSUPPORTED;
    if (fused) return nsvd_fused_features(*desc, *params, *prob, x, B, ws, save_for_backward & 1, (hipStream_t)stream);
    if (!(prob->eps > 0.f)) return NSVD_EUNSUPPORTED;
    const GenericWs w = carve(*desc, B, ws);
    const int E = 1 + 2 * desc->D;
    return nsvd_fourier_features_evenodd(x, params->fourier_B, w.phiT, B, desc->D, desc->m, prob->eps, E * B,
                                         (hipStream_t)stream);
}

namespace {
int sample_features_impl(const nsvd_model_desc* desc, const nsvd_params* params, const nsvd_problem* prob,
                         unsigned long long seed, unsigned long long offset, const nsvd_step_state* state, float* x,
                         int B, void* ws, size_t ws_bytes, int save_for_backward, int path, void* stream) {
    int rc = validate(desc);
    if (rc) return rc;
    if (!prob || !x || !ws || !params || !params->fourier_B || B <= 0) return NSVD_EINVAL;
    if (ws_bytes < nsvd_workspace_bytes(desc, B)) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    const bool fused = want_fused(*desc, B, path, !(prob->eps > 0.f));
    if ((path == NSVD_PATH_FUSED || path == NSVD_PATH_FUSED_BF16X3) && !fused) return NSVD_EUNSUPPORTED;
    NsvdSampler smp;
    smp.seed = seed;
    smp.offset = offset;
    smp.sigma = prob->sigma;
    smp.on = 1;
    smp.offset_add = state ? (const unsigned long long*)&state->step : nullptr;
    hipStream_t s = (hipStream_t)stream;
    if (fused) return nsvd_fused_features(*desc, *params, *prob, x, B, ws, save_for_backward & 1, s, &smp, x);
    if (!(prob->eps > 0.f)) return NSVD_EUNSUPPORTED;
    rc = nsvd_sample_launch(smp, x, B, desc->D, s);
    if (rc) return rc;
    const GenericWs w = carve(*desc, B, ws);
    const int E = 1 + 2 * desc->D;
    return nsvd_fourier_features_evenodd(x, params->fourier_B, w.phiT, B, desc->D, desc->m, prob->eps, E * B,
                                         (hipStream_t)stream);
}
}  // namespace

extern "C" int nsvd_operator_sample_features(const nsvd_model_desc* desc, const nsvd_params* params,
                                             const nsvd_problem* prob, unsigned long long seed,
                                             unsigned long long offset, float* x, int B, void* ws, size_t ws_bytes,
                                             int save_for_backward, int path, void* stream) {
    return sample_features_impl(desc, params, prob, seed, offset, nullptr, x, B, ws, ws_bytes, save_for_backward, path,
                                stream);
}

extern "C" int nsvd_operator_sample_features_dev(const nsvd_model_desc* desc, const nsvd_params* params,
                                                 const nsvd_problem* prob, unsigned long long seed,
                                                 unsigned long long offset_base, const nsvd_step_state* state,
                                                 float* x, int B, void* ws, size_t ws_bytes, int save_for_backward,
                                                 int path, void* stream) {
    if (!state || ((uintptr_t)state & 7) != 0) return NSVD_EINVAL;
    return sample_features_impl(desc, params, prob, seed, offset_base, state, x, B, ws, ws_bytes, save_for_backward,
                                path, stream);
}

extern "C" int nsvd_model_forward(const nsvd_model_desc* desc, const nsvd_params* params, const float* x, int B,
                                  float hard_mul_const, float* out, void* ws, size_t ws_bytes, int save_for_backward,
                                  void* stream) {
    // plain model evaluation always takes the generic kernels, on a workspace laid out for the centre rows only
    int rc = validate(desc, MODEL_MAX_D);
    if (rc) return rc;
    if (!x || !out || !ws || B <= 0) return NSVD_EINVAL;
    rc = check_params(*desc, params, true);
    if (rc) return rc;
    if (ws_bytes < nsvd_model_workspace_bytes(desc, B)) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    if (nsvd_fused_model_supported(*desc, B))  // 128-wide hidden layers: the E = 1 instance of the MFMA forward
        return nsvd_fused_model_forward(*desc, *params, x, B, hard_mul_const, out, ws, save_for_backward,
                                        (hipStream_t)stream);
    const GenericWs w = carve(*desc, B, ws, 1);
    const int R = w.R;
    rc = generic_mlp(*desc, *params, x, B, 0.f, 1, w, (hipStream_t)stream);
    if (rc) return rc;
    return nsvd_model_out(w.z[desc->nlayers - 1], R, x, desc->has_exp_mask ? params->scales : nullptr,
                          hard_mul_const, B, desc->D, desc->L, out, save_for_backward ? w.jac : nullptr,
                          (save_for_backward && desc->has_exp_mask) ? w.dsc : nullptr, (hipStream_t)stream);
}

extern "C" int nsvd_model_backward(const nsvd_model_desc* desc, const nsvd_params* params, const float* x, int B,
                                   const float* dout, const nsvd_params* grads, void* ws, size_t ws_bytes,
                                   void* stream) {
    int rc = validate(desc, MODEL_MAX_D);
    if (rc) return rc;
    if (!x || !dout || !ws || B <= 0) return NSVD_EINVAL;
    rc = check_params(*desc, params, true);
    if (rc) return rc;
    rc = check_params(*desc, grads, false);
    if (rc) return rc;
    if (ws_bytes < nsvd_model_workspace_bytes(desc, B)) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    if (nsvd_fused_model_supported(*desc, B)) {
        nsvd_problem unused;
        memset(&unused, 0, sizeof(unused));
        return nsvd_fused_backward(*desc, *params, unused, x, B, dout, *grads, ws, (hipStream_t)stream);
    }
    return generic_backward(*desc, *params, x, B, dout, *grads, ws, (hipStream_t)stream, 1);
}

extern "C" int nsvd_operator_backward(const nsvd_model_desc* desc, const nsvd_params* params,
                                      const nsvd_problem* prob, const float* x, int B, const float* df,
                                      const nsvd_params* grads, void* ws, size_t ws_bytes, int path, void* stream) {
    int rc = validate(desc);
    if (rc) return rc;
    if (!prob || !x || !df || !ws || B <= 0) return NSVD_EINVAL;
    rc = check_params(*desc, params, true);
    if (rc) return rc;
    rc = check_params(*desc, grads, false);
    if (rc) return rc;
    if (ws_bytes < nsvd_workspace_bytes(desc, B)) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    const bool fused = want_fused(*desc, B, path, !(prob->eps > 0.f));
    if ((path == NSVD_PATH_FUSED || path == NSVD_PATH_FUSED_BF16X3) && !fused) return NSVD_EUNSUPPORTED;
    if (fused) return nsvd_fused_backward(*desc, *params, *prob, x, B, df, *grads, ws, (hipStream_t)stream);
    return generic_backward(*desc, *params, x, B, df, *grads, ws, (hipStream_t)stream);
}

namespace {
int backward_evd_impl(const nsvd_model_desc* desc, const nsvd_params* params, const nsvd_problem* prob,
                      const float* x, int B, const float* f, const float* Tf, int mask_kind, const float* v,
                      const float* M, float* moments, int moments_reduced, const void* evd_scratch, int L_total,
                      int l_offset, float grad_scale, float* loss, const nsvd_params* grads,
                      const nsvd_rmsprop* opt, void* ws, size_t ws_bytes, int path, void* stream, int l_begin = 0,
                      int l_count = 0, const NsvdNextBatch* next = nullptr, bool model_mode = false,
                      int window_of_step = 0, int not_last = 0, void* ev_after_chain = nullptr) {
    // model_mode: the forward was nsvd_model_forward (plain model evaluation, any input dimension up to 64, no
    // Hamiltonian): Tf is whatever operator output the caller computed from f (the kernel-operator path)
    int rc = validate(desc, model_mode ? MODEL_MAX_D : 4);
    if (rc) return rc;
    if ((!prob && !model_mode) || !x || !f || !Tf || !ws || B <= 0) return NSVD_EINVAL;
    // direct mode (fused path): neither reduced moments nor partial sums - the backward kernel takes the moments it
    // needs from f itself; `moments` and `loss` are then not written
    const bool direct = !moments_reduced && !evd_scratch;
    if (!direct && !moments) return NSVD_EINVAL;
    if (mask_kind < 0 || mask_kind > NSVD_MASK_JOINT) return NSVD_EINVAL;
    if (mask_kind == NSVD_MASK_CUSTOM && (!v || !M)) return NSVD_EINVAL;
    rc = check_params(*desc, params, true);
    if (rc) return rc;
    if (grads || !opt) {
        rc = check_params(*desc, grads, false);
        if (rc) return rc;
    }
    NsvdOptStep st;
    memset(&st, 0, sizeof(st));
    if (opt) {
        rc = check_params(*desc, &opt->sq, false);
        if (rc) return rc;
        if (opt->has_ema) {
            rc = check_params(*desc, &opt->ema, false);
            if (rc) return rc;
            st.ema = &opt->ema;
        }
        st.sq = opt->sq;
        st.h = nsvd_make_hyper(opt->lr, opt->alpha, opt->eps, opt->has_ema ? opt->ema_decay : 0.0, 1.0);
        st.state = opt->state;
        if (st.state && ((uintptr_t)st.state & 7) != 0) return NSVD_EINVAL;
        // (NSVD_PLANES_FROM_STEP=0: measurements of the split launch against the epilogue's emission)
        static const bool planes_env = [] { const char* e = getenv("NSVD_PLANES_FROM_STEP"); return !e || atoi(e) != 0; }();
        st.emit_planes = !model_mode && path == NSVD_PATH_FUSED_BF16X3 && planes_env;
    }
    if (ws_bytes < (model_mode ? nsvd_model_workspace_bytes(desc, B) : nsvd_workspace_bytes(desc, B))) return NSVD_EINVAL;
    if (((uintptr_t)ws & 255) != 0) return NSVD_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const bool fused = model_mode ? nsvd_fused_model_supported(*desc, B) : want_fused(*desc, B, path, !(prob->eps > 0.f));
    if (model_mode && !fused) return NSVD_EUNSUPPORTED;  // plain-model training steps exist on the MFMA kernels only
    if ((path == NSVD_PATH_FUSED || path == NSVD_PATH_FUSED_BF16X3) && !fused) return NSVD_EUNSUPPORTED;
    if (L_total <= 0) L_total = desc->L;
    if (l_offset < 0 || l_offset + desc->L > L_total || L_total > 128) return NSVD_EINVAL;
    const int L = L_total;  // f, Tf, moments and masks are indexed by GLOBAL head
    const NsvdEvdChunking c = nsvd_evd_chunking(B);
    if (fused) {
        NsvdEvdIn in;
        memset(&in, 0, sizeof(in));
        in.f = f; in.Tf = Tf; in.v = v; in.M = M;
        in.kind = mask_kind;
        in.grad_scale = grad_scale;
        in.loss = loss;
        in.Lg = L_total;
        in.l_off = l_offset;
        if (direct) {
            // (the loss scalars come from per-head partials of the chain kernel, added by the weight-gradient kernel)
        } else if (moments_reduced) {
            in.moments = moments;
        } else {
            in.part = (const float*)evd_scratch;
            in.part_op = in.part + (size_t)(c.n1 + c.n2) * L * L;
            in.moments_out = moments;
        }
        if (l_count < 0 || l_begin < 0 || l_begin + l_count > desc->L) return NSVD_EINVAL;
        return nsvd_fused_backward_evd(*desc, *params, B, in, grads, opt ? &st : nullptr, ws, s, l_begin, l_count,
                                       next, window_of_step, not_last, ev_after_chain);
    }
    if (window_of_step) return NSVD_EUNSUPPORTED;  // windows of a fused step exist on the fused kernels only
    if (next) return NSVD_EUNSUPPORTED;  // guest feature workgroups exist on the fused kernels only
    if (opt && opt->state) return NSVD_EUNSUPPORTED;  // the device-resident schedule is read by the fused kernels only
    if (l_count > 0 && l_count != desc->L) return NSVD_EUNSUPPORTED;  // head windows need the fused kernels
    // generic path: finish the loss with the stand-alone kernels, then the layer-by-layer backward
    if (direct) return NSVD_EINVAL;  // needs the partial moments (evd_scratch) or the reduced ones
    if (L_total != desc->L) return NSVD_EUNSUPPORTED;  // head-parallel sharding needs the fused kernels
    const GenericWs w = carve(*desc, B, ws);
    if (!moments_reduced) {
        rc = nsvd_evd_reduce_partials(evd_scratch, B, L, moments, s);
        if (rc) return rc;
    }
    rc = nsvd_evd_loss_grad(f, Tf, B, L, mask_kind, v, M, moments, grad_scale, loss, w.df, stream);
    if (rc) return rc;
    if (!grads) return NSVD_EINVAL;  // the generic path needs somewhere to put the gradients
    rc = generic_backward(*desc, *params, x, B, w.df, *grads, ws, s);
    if (rc || !opt) return rc;
    // generic path + fused-step request: ONE stand-alone optimiser launch over the table of tensors (unaligned tensors -
    // never with torch allocations -: one launch per tensor)
    const int F = 2 * desc->m;
    {
        NsvdOptTable tab;
        memset(&tab, 0, sizeof(tab));
        int c = 0;
        auto add = [&](float* pp, const float* gg, float* qq, float* ee, size_t n) {
            tab.p[c] = pp; tab.g[c] = gg; tab.sq[c] = qq; tab.ema[c] = ee; tab.n[c] = n;
            ++c;
        };
        for (int i = 0; i < desc->nlayers; ++i) {
            const size_t hin = i == 0 ? (size_t)F : (size_t)desc->dims[i - 1], hout = (size_t)desc->dims[i];
            add(params->W[i], grads->W[i], st.sq.W[i], st.ema ? st.ema->W[i] : nullptr, (size_t)desc->L * hout * hin);
            add(params->b[i], grads->b[i], st.sq.b[i], st.ema ? st.ema->b[i] : nullptr, (size_t)desc->L * hout);
        }
        if (desc->has_exp_mask)
            add(params->scales, grads->scales, st.sq.scales, st.ema ? st.ema->scales : nullptr, (size_t)desc->L);
        tab.count = c;
        static const char* et = getenv("NSVD_OPT_TABLE");  // 0: one launch per tensor (A/B, bit-identical)
        rc = (et && et[0] == '0') ? NSVD_EUNSUPPORTED : nsvd_rmsprop_table_launch(tab, st.h, s);
        if (rc != NSVD_EUNSUPPORTED) return rc;
    }
    for (int i = 0; i < desc->nlayers; ++i) {
        const size_t hin = i == 0 ? (size_t)F : (size_t)desc->dims[i - 1], hout = (size_t)desc->dims[i];
        rc = nsvd_rmsprop_launch(params->W[i], grads->W[i], st.sq.W[i], st.ema ? st.ema->W[i] : nullptr,
                                 (size_t)desc->L * hout * hin, st.h, s);
        if (rc) return rc;
        rc = nsvd_rmsprop_launch(params->b[i], grads->b[i], st.sq.b[i], st.ema ? st.ema->b[i] : nullptr,
                                 (size_t)desc->L * hout, st.h, s);
        if (rc) return rc;
    }
    if (desc->has_exp_mask)
        rc = nsvd_rmsprop_launch(params->scales, grads->scales, st.sq.scales, st.ema ? st.ema->scales : nullptr,
                                 (size_t)desc->L, st.h, s);
    return rc;
}
}  // namespace

extern "C" int nsvd_operator_backward_evd(const nsvd_model_desc* desc, const nsvd_params* params,
                                          const nsvd_problem* prob, const float* x, int B, const float* f,
                                          const float* Tf, int mask_kind, const float* v, const float* M,
                                          float* moments, int moments_reduced, const void* evd_scratch,
                                          int L_total, int l_offset, float grad_scale, float* loss,
                                          const nsvd_params* grads, void* ws, size_t ws_bytes, int path,
                                          void* stream) {
    if (!grads) return NSVD_EINVAL;
    return backward_evd_impl(desc, params, prob, x, B, f, Tf, mask_kind, v, M, moments, moments_reduced,
                             evd_scratch, L_total, l_offset, grad_scale, loss, grads, nullptr, ws, ws_bytes, path,
                             stream);
}

extern "C" int nsvd_operator_backward_evd_heads(const nsvd_model_desc* desc, const nsvd_params* params,
                                                const nsvd_problem* prob, const float* x, int B, const float* f,
                                                const float* Tf, int mask_kind, const float* v, const float* M,
                                                float* moments, int moments_reduced, const void* evd_scratch,
                                                int L_total, int l_offset, float grad_scale, float* loss,
                                                const nsvd_params* grads, void* ws, size_t ws_bytes, int path,
                                                int l_begin, int l_count, void* stream) {
    if (!grads || l_count <= 0) return NSVD_EINVAL;
    return backward_evd_impl(desc, params, prob, x, B, f, Tf, mask_kind, v, M, moments, moments_reduced,
                             evd_scratch, L_total, l_offset, grad_scale, loss, grads, nullptr, ws, ws_bytes, path,
                             stream, l_begin, l_count);
}

extern "C" int nsvd_backward_head_window_ok(const nsvd_model_desc* desc, const nsvd_problem* prob, int B, int path,
                                            int l_count) {
    if (validate(desc) != 0 || !prob || B <= 0) return 0;
    if (!want_fused(*desc, B, path, !(prob->eps > 0.f))) return l_count == desc->L;
    return nsvd_fused_backward_window_ok(*desc, B, l_count) ? 1 : 0;
}

extern "C" int nsvd_operator_backward_evd_step(const nsvd_model_desc* desc, const nsvd_params* params,
                                               const nsvd_problem* prob, const float* x, int B, const float* f,
                                               const float* Tf, int mask_kind, const float* v, const float* M,
                                               float* moments, int moments_reduced, const void* evd_scratch,
                                               int L_total, int l_offset, float grad_scale, float* loss,
                                               const nsvd_params* grads, const nsvd_rmsprop* opt, void* ws,
                                               size_t ws_bytes, int path, void* stream) {
    if (!opt) return NSVD_EINVAL;
    return backward_evd_impl(desc, params, prob, x, B, f, Tf, mask_kind, v, M, moments, moments_reduced,
                             evd_scratch, L_total, l_offset, grad_scale, loss, grads, opt, ws, ws_bytes, path,
                             stream);
}

extern "C" int nsvd_operator_backward_evd_step_window(const nsvd_model_desc* desc, const nsvd_params* params,
                                                      const nsvd_problem* prob, const float* x, int B, const float* f,
                                                      const float* Tf, int mask_kind, const float* v, const float* M,
                                                      float* moments, int moments_reduced, const void* evd_scratch,
                                                      int L_total, int l_offset, float grad_scale, float* loss,
                                                      const nsvd_params* grads, const nsvd_rmsprop* opt, void* ws,
                                                      size_t ws_bytes, int path, int l_begin, int l_count,
                                                      int last_window, void* ev_after_chain,
                                                      unsigned long long next_seed, unsigned long long next_offset,
                                                      float* x_next, void* ws_next, size_t ws_next_bytes,
                                                      void* stream) {
    if (!opt || !prob || l_count <= 0) return NSVD_EINVAL;
    NsvdNextBatch nb;
    if (x_next) {  // the next batch rides in THIS window's chain launch (pass it to one window of the step only)
        if (!ws_next || ws_next == ws) return NSVD_EINVAL;
        if (ws_next_bytes < nsvd_workspace_bytes(desc, B) || ((uintptr_t)ws_next & 255) != 0) return NSVD_EINVAL;
        memset(&nb, 0, sizeof(nb));
        nb.smp.seed = next_seed;
        nb.smp.offset = next_offset;
        nb.smp.sigma = prob->sigma;
        nb.smp.on = 1;
        nb.x = x_next;
        nb.ws = ws_next;
        nb.eps = prob->eps;
    }
    return backward_evd_impl(desc, params, prob, x, B, f, Tf, mask_kind, v, M, moments, moments_reduced,
                             evd_scratch, L_total, l_offset, grad_scale, loss, grads, opt, ws, ws_bytes, path,
                             stream, l_begin, l_count, x_next ? &nb : nullptr, false, 1, last_window ? 0 : 1,
                             ev_after_chain);
}

extern "C" int nsvd_operator_backward_evd_step_next(const nsvd_model_desc* desc, const nsvd_params* params,
                                                    const nsvd_problem* prob, const float* x, int B, const float* f,
                                                    const float* Tf, int mask_kind, const float* v, const float* M,
                                                    float* moments, int moments_reduced, const void* evd_scratch,
                                                    int L_total, int l_offset, float grad_scale, float* loss,
                                                    const nsvd_params* grads, const nsvd_rmsprop* opt, void* ws,
                                                    size_t ws_bytes, int path, unsigned long long next_seed,
                                                    unsigned long long next_offset, float* x_next, void* ws_next,
                                                    size_t ws_next_bytes, void* stream) {
    if (!opt || !prob || !x_next || !ws_next || ws_next == ws) return NSVD_EINVAL;
    if (ws_next_bytes < nsvd_workspace_bytes(desc, B) || ((uintptr_t)ws_next & 255) != 0) return NSVD_EINVAL;
    NsvdNextBatch nb;
    memset(&nb, 0, sizeof(nb));
    nb.smp.seed = next_seed;
    nb.smp.offset = next_offset;
    nb.smp.sigma = prob->sigma;
    nb.smp.on = 1;
    nb.x = x_next;
    nb.ws = ws_next;
    nb.eps = prob->eps;
    return backward_evd_impl(desc, params, prob, x, B, f, Tf, mask_kind, v, M, moments, moments_reduced,
                             evd_scratch, L_total, l_offset, grad_scale, loss, grads, opt, ws, ws_bytes, path,
                             stream, 0, 0, &nb);
}

extern "C" int nsvd_model_backward_evd_step(const nsvd_model_desc* desc, const nsvd_params* params, const float* x,
                                            int B, const float* f, const float* Tf, int mask_kind, const float* v,
                                            const float* M, float* moments, int moments_reduced,
                                            const void* evd_scratch, int L_total, int l_offset, float grad_scale,
                                            float* loss, const nsvd_params* grads, const nsvd_rmsprop* opt, void* ws,
                                            size_t ws_bytes, void* stream) {
    if (!opt && !grads) return NSVD_EINVAL;
    return backward_evd_impl(desc, params, nullptr, x, B, f, Tf, mask_kind, v, M, moments, moments_reduced, evd_scratch,
                             L_total, l_offset, grad_scale, loss, grads, opt, ws, ws_bytes, NSVD_PATH_AUTO, stream, 0, 0,
                             nullptr, true);
}
