// torch.utils.cpp_extension-style binding of the hot-path entry points of include/nsvd.h (BASELINE.json north_star: "a
// thin torch.utils.cpp_extension HIP module exposing the same methods/ loss-function signature"): tensors in, C ABI
// out. No kernel lives here - this file is host C++ compiled by the host compiler against the torch headers and linked
// to libnsvd_hip.so; every function
//   * checks its tensors in C++ (GPU, float32, contiguous, one device, shapes against the model description),
//   * takes the CURRENT HIP stream of that device (c10::hip::getCurrentHIPStream),
//   * forwards to the C ABI and turns a non-zero return code into a c10::Error (TORCH_CHECK), as SURVEY 8(b) asks.
// Selected with NSVD_BINDING=torch (neural_svd_amd/hip_ops.py); the default binding is ctypes (neural_svd_amd/_lib.py:
// no compile step beyond the HIP library, and - measured, DESIGN.md 1 - not slower per call). Same entry points, same
// structs: tests/test_abi.py checks both against the header.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>

#include <string>
#include <vector>

#include "../../include/nsvd.h"

namespace {

void check_rc(int rc, const char* what) {
    if (rc == 0) return;
    if (rc == NSVD_EINVAL) TORCH_CHECK(false, what, ": invalid argument (NSVD_EINVAL)");
    if (rc == NSVD_EUNSUPPORTED) TORCH_CHECK(false, what, ": unsupported configuration (NSVD_EUNSUPPORTED)");
    TORCH_CHECK(false, what, ": HIP error ", -rc);
}

// device the call runs on: every tensor argument must live there
struct Dev {
    int index = -1;
    void see(const at::Tensor& t, const char* name) {
        TORCH_CHECK(t.is_cuda(), name, " must live on the GPU (got ", t.device(), "); neural_svd_amd has no CPU path");
        if (index < 0) index = t.get_device();
        TORCH_CHECK(t.get_device() == index, name, ": tensors on different devices");
    }
    // (call LAST: the device is known once every tensor has been seen - the wrappers below fetch their pointers into
    // locals first, C++ leaves the evaluation order of call arguments open)
    void* stream() const {
        TORCH_CHECK(index >= 0, "no GPU tensor among the arguments");
        return (void*)c10::hip::getCurrentHIPStream(index).stream();
    }
};

float* f32(Dev& d, const at::Tensor& t, const char* name) {
    d.see(t, name);
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32 (got ", t.scalar_type(), ")");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    return t.data_ptr<float>();
}
float* f32_opt(Dev& d, const c10::optional<at::Tensor>& t, const char* name) {
    return t.has_value() ? f32(d, *t, name) : nullptr;
}
void* bytes(Dev& d, const at::Tensor& t, const char* name) {
    d.see(t, name);
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    return t.data_ptr();
}

// WaveFunctions(ParallelMLP(FourierFeatures)) shape: examples/operator/pde/__init__.py:19-55
struct Shape {
    nsvd_model_desc d;
    Shape(int L, int D, int m, std::vector<int> dims, bool has_exp_mask) {
        memset(&d, 0, sizeof(d));
        TORCH_CHECK(dims.size() >= 1 && dims.size() <= NSVD_MAX_LAYERS, "at most ", NSVD_MAX_LAYERS, " layers");
        d.L = L; d.D = D; d.m = m; d.nlayers = (int)dims.size();
        for (size_t i = 0; i < dims.size(); ++i) d.dims[i] = dims[i];
        d.has_exp_mask = has_exp_mask ? 1 : 0;
    }
};

// one parameter set (weights, gradients, RMSprop state or EMA shadow): nsvd_params + the tensors it points into
struct ParamSet {
    nsvd_params p;
    std::vector<at::Tensor> keep;
    int device = -1;
    ParamSet(const Shape& s, std::vector<at::Tensor> W, std::vector<at::Tensor> b, c10::optional<at::Tensor> fourier_B,
             c10::optional<at::Tensor> scales) {
        memset(&p, 0, sizeof(p));
        Dev dv;
        const nsvd_model_desc& d = s.d;
        TORCH_CHECK((int)W.size() == d.nlayers && (int)b.size() == d.nlayers,
                    "number of weight / bias tensors does not match the model shape");
        if (fourier_B.has_value()) {
            TORCH_CHECK(fourier_B->dim() == 2 && fourier_B->size(0) == d.D && fourier_B->size(1) == d.m,
                        "fourier_B must be (D, m)");
            p.fourier_B = f32(dv, *fourier_B, "fourier_B");
            keep.push_back(*fourier_B);
        }
        int64_t prev = 2 * (int64_t)d.m;
        for (int i = 0; i < d.nlayers; ++i) {
            TORCH_CHECK(W[i].dim() == 3 && W[i].size(0) == d.L && W[i].size(1) == d.dims[i] && W[i].size(2) == prev,
                        "W[", i, "] must be (L, h_i, h_{i-1})");
            TORCH_CHECK(b[i].numel() == (int64_t)d.L * d.dims[i], "b[", i, "] must have L * h_i elements");
            p.W[i] = f32(dv, W[i], "W[i]");
            p.b[i] = f32(dv, b[i], "b[i]");
            keep.push_back(W[i]);
            keep.push_back(b[i]);
            prev = d.dims[i];
        }
        if (d.has_exp_mask) {
            TORCH_CHECK(scales.has_value() && scales->numel() == d.L, "scales (L,) required when has_exp_mask");
            p.scales = f32(dv, *scales, "scales");
            keep.push_back(*scales);
        }
        device = dv.index;
    }
};

// OperatorWrapper(NegativeHamiltonian(potential), scale, shift) with Gaussian importance: nsvd_problem
struct Problem {
    nsvd_problem q;
    Problem(int potential, double charge_or_k, double eps, double op_scale, double op_shift, double sigma,
            double scale_kinetic, double hard_mul_const, bool use_importance) {
        memset(&q, 0, sizeof(q));
        q.potential = potential;
        q.charge_or_k = (float)charge_or_k;
        q.scale_kinetic = (float)scale_kinetic;
        q.eps = (float)eps;
        q.op_scale = (float)op_scale;
        q.op_shift = (float)op_shift;
        q.sigma = (float)sigma;
        q.hard_mul_const = (float)hard_mul_const;
        q.use_importance = use_importance ? 1 : 0;
    }
};

void see_params(Dev& dv, const ParamSet& ps) {
    if (ps.device < 0) return;
    if (dv.index < 0) dv.index = ps.device;
    TORCH_CHECK(ps.device == dv.index, "parameter set on another device than the call's tensors");
}

int64_t workspace_bytes(const Shape& s, int64_t B) { return (int64_t)nsvd_workspace_bytes(&s.d, (int)B); }

std::string path_name(const Shape& s, const Problem* prob, int64_t B, int64_t path) {
    return prob ? nsvd_path_name_for(&s.d, &prob->q, (int)B, (int)path) : nsvd_path_name(&s.d, (int)B, (int)path);
}

// Tf, f = operator(method, x, importance): examples/__init__.py:7-9 -> ... -> models/mlp.py:204-221
void operator_forward(const Shape& s, const ParamSet& params, const Problem& prob, const at::Tensor& x, at::Tensor f,
                      at::Tensor Tf, at::Tensor ws, int64_t save_flags, int64_t path) {
    Dev dv;
    see_params(dv, params);
    TORCH_CHECK(x.dim() == 2 && x.size(1) == s.d.D, "x must be (B, D)");
    const int B = (int)x.size(0);
    TORCH_CHECK(f.dim() == 2 && f.size(0) == B && f.size(1) == s.d.L && Tf.sizes() == f.sizes(), "f, Tf must be (B, L)");
    float *xp = f32(dv, x, "x"), *fp = f32(dv, f, "f"), *tp = f32(dv, Tf, "Tf");
    void* wp = bytes(dv, ws, "ws");
    check_rc(nsvd_operator_forward(&s.d, &params.p, &prob.q, xp, B, fp, tp, wp, (size_t)ws.numel() * ws.element_size(),
                                   (int)save_flags, (int)path, dv.stream()),
             "nsvd_operator_forward");
}

void operator_backward(const Shape& s, const ParamSet& params, const Problem& prob, const at::Tensor& x,
                       const at::Tensor& df, const ParamSet& grads, at::Tensor ws, int64_t path) {
    Dev dv;
    see_params(dv, params);
    see_params(dv, grads);
    const int B = (int)x.size(0);
    TORCH_CHECK(df.dim() == 2 && df.size(0) == B && df.size(1) == s.d.L, "df must be (B, L)");
    float *xp = f32(dv, x, "x"), *dp = f32(dv, df, "df");
    void* wp = bytes(dv, ws, "ws");
    check_rc(nsvd_operator_backward(&s.d, &params.p, &prob.q, xp, B, dp, &grads.p, wp,
                                    (size_t)ws.numel() * ws.element_size(), (int)path, dv.stream()),
             "nsvd_operator_backward");
}

void operator_sample_features(const Shape& s, const ParamSet& params, const Problem& prob, uint64_t seed,
                              uint64_t offset, c10::optional<at::Tensor> state, at::Tensor x, at::Tensor ws,
                              bool save_for_backward, int64_t path) {
    Dev dv;
    see_params(dv, params);
    const int B = (int)x.size(0);
    float* xp = f32(dv, x, "x");
    void* wp = bytes(dv, ws, "ws");
    const nsvd_step_state* st = state.has_value() ? (const nsvd_step_state*)bytes(dv, *state, "state") : nullptr;
    const size_t wn = (size_t)ws.numel() * ws.element_size();
    if (st)
        check_rc(nsvd_operator_sample_features_dev(&s.d, &params.p, &prob.q, seed, offset, st, xp, B, wp, wn,
                                                   save_for_backward ? 1 : 0, (int)path, dv.stream()),
                 "nsvd_operator_sample_features_dev");
    else
        check_rc(nsvd_operator_sample_features(&s.d, &params.p, &prob.q, seed, offset, xp, B, wp, wn,
                                               save_for_backward ? 1 : 0, (int)path, dv.stream()),
                 "nsvd_operator_sample_features");
}

// NestedLoRALossFunctionEVD.forward / backward pieces: methods/nestedlora.py:70-111
void evd_moments(const at::Tensor& f, const at::Tensor& Tf, int64_t mask_kind, c10::optional<at::Tensor> v,
                 at::Tensor moments, at::Tensor scratch) {
    Dev dv;
    const int B = (int)f.size(0), L = (int)f.size(1);
    TORCH_CHECK(Tf.sizes() == f.sizes() && moments.numel() == 2 * (int64_t)L * L + 1, "f, Tf (B, L); moments 2 L^2 + 1");
    float *fp = f32(dv, f, "f"), *tp = f32(dv, Tf, "Tf"), *vp = f32_opt(dv, v, "v"), *mp = f32(dv, moments, "moments");
    void* sp = bytes(dv, scratch, "scratch");
    check_rc(nsvd_evd_moments(fp, tp, B, L, (int)mask_kind, vp, mp, sp, dv.stream()), "nsvd_evd_moments");
}

void evd_loss_grad(const at::Tensor& f, const at::Tensor& Tf, int64_t mask_kind, c10::optional<at::Tensor> v,
                   c10::optional<at::Tensor> M, const at::Tensor& moments, double grad_scale, at::Tensor loss,
                   c10::optional<at::Tensor> df) {
    Dev dv;
    const int B = (int)f.size(0), L = (int)f.size(1);
    TORCH_CHECK(Tf.sizes() == f.sizes() && loss.numel() >= 3, "f, Tf (B, L); loss (3)");
    float *fp = f32(dv, f, "f"), *tp = f32(dv, Tf, "Tf"), *vp = f32_opt(dv, v, "v"), *Mp = f32_opt(dv, M, "M");
    float *mp = f32(dv, moments, "moments"), *lp = f32(dv, loss, "loss"), *dp = f32_opt(dv, df, "df");
    check_rc(nsvd_evd_loss_grad(fp, tp, B, L, (int)mask_kind, vp, Mp, mp, (float)grad_scale, lp, dp, dv.stream()),
             "nsvd_evd_loss_grad");
}

struct Rmsprop {
    nsvd_rmsprop o;
    Rmsprop(const ParamSet& sq, const ParamSet* ema, double lr, double alpha, double eps, double ema_decay,
            c10::optional<at::Tensor> state) {
        memset(&o, 0, sizeof(o));
        o.sq = sq.p;
        if (ema) o.ema = ema->p;
        o.lr = lr; o.alpha = alpha; o.eps = eps; o.ema_decay = ema_decay;
        o.has_ema = ema ? 1 : 0;
        o.state = state.has_value() ? (nsvd_step_state*)state->data_ptr() : nullptr;
    }
};

// loss.backward(); optimizer.step(); ema.update() of examples/operator/__init__.py:68-73 in the backward's kernels
void operator_backward_evd_step(const Shape& s, const ParamSet& params, const Problem& prob, const at::Tensor& x,
                                const at::Tensor& f, const at::Tensor& Tf, int64_t mask_kind,
                                c10::optional<at::Tensor> v, c10::optional<at::Tensor> M,
                                c10::optional<at::Tensor> moments, bool moments_reduced,
                                c10::optional<at::Tensor> evd_scratch, int64_t l_offset, double grad_scale,
                                c10::optional<at::Tensor> loss, const ParamSet* grads, const Rmsprop& opt, at::Tensor ws,
                                int64_t path, c10::optional<at::Tensor> x_next, c10::optional<at::Tensor> ws_next,
                                uint64_t next_seed, uint64_t next_offset) {
    Dev dv;
    see_params(dv, params);
    const int B = (int)x.size(0), Lt = (int)f.size(1);
    TORCH_CHECK(f.dim() == 2 && f.size(0) == B && Tf.sizes() == f.sizes() && Lt >= s.d.L, "f, Tf must be (B, L_total)");
    void* scr = evd_scratch.has_value() ? bytes(dv, *evd_scratch, "evd_scratch") : nullptr;
    float *xp = f32(dv, x, "x"), *fp = f32(dv, f, "f"), *tp = f32(dv, Tf, "Tf"), *vp = f32_opt(dv, v, "v");
    float *Mp = f32_opt(dv, M, "M"), *mp = f32_opt(dv, moments, "moments"), *lp = f32_opt(dv, loss, "loss");
    void* wp = bytes(dv, ws, "ws");
    const size_t wn = (size_t)ws.numel() * ws.element_size();
    const nsvd_params* gp = grads ? &grads->p : nullptr;
    if (x_next.has_value()) {
        TORCH_CHECK(ws_next.has_value(), "ws_next required with x_next");
        float* xn = f32(dv, *x_next, "x_next");
        void* wnp = bytes(dv, *ws_next, "ws_next");
        check_rc(nsvd_operator_backward_evd_step_next(&s.d, &params.p, &prob.q, xp, B, fp, tp, (int)mask_kind, vp, Mp, mp,
                                                      moments_reduced ? 1 : 0, scr, Lt, (int)l_offset, (float)grad_scale,
                                                      lp, gp, &opt.o, wp, wn, (int)path, next_seed, next_offset, xn, wnp,
                                                      (size_t)ws_next->numel() * ws_next->element_size(), dv.stream()),
                 "nsvd_operator_backward_evd_step_next");
        return;
    }
    check_rc(nsvd_operator_backward_evd_step(&s.d, &params.p, &prob.q, xp, B, fp, tp, (int)mask_kind, vp, Mp, mp,
                                             moments_reduced ? 1 : 0, scr, Lt, (int)l_offset, (float)grad_scale, lp, gp,
                                             &opt.o, wp, wn, (int)path, dv.stream()),
             "nsvd_operator_backward_evd_step");
}

// torch.optim.RMSprop step + torch_ema update: examples/utils.py:50-57, examples/operator/__init__.py:69-73
void rmsprop_ema_step(at::Tensor p, const at::Tensor& grad, at::Tensor sq, c10::optional<at::Tensor> ema, double lr,
                      double alpha, double eps, double ema_decay, double grad_scale) {
    Dev dv;
    const size_t n = (size_t)p.numel();
    TORCH_CHECK((size_t)grad.numel() == n && (size_t)sq.numel() == n && (!ema.has_value() || (size_t)ema->numel() == n),
                "rmsprop_ema_step: size mismatch");
    float *pp = f32(dv, p, "p"), *gp = f32(dv, grad, "grad"), *sp = f32(dv, sq, "sq"), *ep = f32_opt(dv, ema, "ema");
    check_rc(nsvd_rmsprop_ema_step(pp, gp, sp, ep, n, lr, alpha, eps, ema_decay, grad_scale, dv.stream()),
             "nsvd_rmsprop_ema_step");
}

void spectrum_accumulate(const at::Tensor& f, const at::Tensor& Tf, const at::Tensor& x, double sigma,
                         bool use_importance, double lim, at::Tensor cov, at::Tensor quad) {
    Dev dv;
    const int B = (int)f.size(0), L = (int)f.size(1), D = (int)x.size(1);
    float *fp = f32(dv, f, "f"), *tp = f32(dv, Tf, "Tf"), *xp = f32(dv, x, "x"), *cp = f32(dv, cov, "cov");
    float* qp = f32(dv, quad, "quad");
    check_rc(nsvd_spectrum_accumulate(fp, tp, xp, B, L, D, (float)sigma, use_importance ? 1 : 0, (float)lim, cp, qp,
                                      dv.stream()),
             "nsvd_spectrum_accumulate");
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "tensor-level binding of libnsvd_hip.so's hot path (include/nsvd.h)";
    m.def("abi_version", []() { return nsvd_abi_version(); });
    // what tests/test_abi.py compares with the header: sizes / offsets as THIS translation unit's compiler sees them
    m.def("struct_layout", []() {
        return std::vector<int64_t>{(int64_t)sizeof(nsvd_model_desc), (int64_t)sizeof(nsvd_params),
                                    (int64_t)sizeof(nsvd_problem), (int64_t)sizeof(nsvd_rmsprop),
                                    (int64_t)offsetof(nsvd_rmsprop, state), (int64_t)sizeof(nsvd_step_state),
                                    (int64_t)offsetof(nsvd_step_state, cur)};
    });
    m.def("bound_entry_points", []() {
        return std::vector<std::string>{"nsvd_abi_version", "nsvd_workspace_bytes", "nsvd_path_name", "nsvd_path_name_for",
                                        "nsvd_operator_forward", "nsvd_operator_backward",
                                        "nsvd_operator_sample_features", "nsvd_operator_sample_features_dev",
                                        "nsvd_evd_moments", "nsvd_evd_loss_grad", "nsvd_operator_backward_evd_step",
                                        "nsvd_operator_backward_evd_step_next", "nsvd_rmsprop_ema_step",
                                        "nsvd_spectrum_accumulate"};
    });
    py::class_<Shape>(m, "Shape").def(py::init<int, int, int, std::vector<int>, bool>());
    py::class_<ParamSet>(m, "ParamSet")
        .def(py::init<const Shape&, std::vector<at::Tensor>, std::vector<at::Tensor>, c10::optional<at::Tensor>,
                      c10::optional<at::Tensor>>());
    py::class_<Problem>(m, "Problem").def(py::init<int, double, double, double, double, double, double, double, bool>());
    py::class_<Rmsprop>(m, "Rmsprop")
        .def(py::init<const ParamSet&, const ParamSet*, double, double, double, double, c10::optional<at::Tensor>>(),
             py::keep_alive<1, 2>(), py::keep_alive<1, 3>());
    m.def("workspace_bytes", &workspace_bytes);
    m.def("path_name", &path_name);
    m.def("operator_forward", &operator_forward);
    m.def("operator_backward", &operator_backward);
    m.def("operator_sample_features", &operator_sample_features);
    m.def("evd_moments", &evd_moments);
    m.def("evd_loss_grad", &evd_loss_grad);
    m.def("operator_backward_evd_step", &operator_backward_evd_step);
    m.def("rmsprop_ema_step", &rmsprop_ema_step);
    m.def("spectrum_accumulate", &spectrum_accumulate);
}
