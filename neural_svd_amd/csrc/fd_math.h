// Per-(sample, head) math of the importance-weighted finite-difference Hamiltonian, shared by the
// generic epilogue kernel and the fused MFMA kernel's epilogue so both paths are the same arithmetic.
#pragma once
#include "nsvd_common.h"

#define NSVD_FD_MAXD 4

struct NsvdFdOut {
    float f, Tf, jac, dsc;
};

// bv[e]: raw head output base_l(x_e) at the stencil points (e = 0 centre, 1+2i: +eps e_i, 2+2i: -eps e_i)
// xc: centre coordinates. Follows the reference's operation order:
//   g_e   = sqrt(p(x_e)) * (c * base_e * mask_l(x_e))          pde/__init__.py:16, diff_ops.py:13
//   lap_g = (-2 D g_0 + sum_i (g_+i + g_-i)) / eps^2           diff_ops.py:38-48
//   lap   = lap_g / clamp(sqrt p(x_0), 1e-5),  fs = g_0 / clamp(...)   diff_ops.py:15-18
//   Tf    = scale * -( -c_k lap + V(x) fs ) + shift * fs       schrodinger/__init__.py:18-22, examples/__init__.py:9
// one stencil point: g_e and (needed of the centre only) sqrt p, mask, |x|
struct NsvdFdG {
    float g, sp, mk, r;
};
__device__ __forceinline__ NsvdFdG nsvd_fd_g(int e, float bve, const float* xc, int D, bool has_mask, float s_l,
                                             const nsvd_problem& prob, float log_norm) {
    float xe[NSVD_FD_MAXD];
    float r2 = 0.f;
    for (int d = 0; d < D; ++d) {
        xe[d] = nsvd_stencil_coord(xc[d], d, e, prob.eps);
        r2 = fmaf(xe[d], xe[d], r2);
    }
    NsvdFdG o;
    o.sp = prob.use_importance ? nsvd_sqrt_gauss_pdf(xe, D, prob.sigma, log_norm) : 1.f;
    float model = prob.hard_mul_const * bve;
    o.mk = 1.f;
    o.r = sqrtf(r2);
    if (has_mask) {
        o.mk = expf(-o.r / s_l);
        model *= o.mk;
    }
    o.g = o.sp * model;
    return o;
}

// stencil combination; g[e] from nsvd_fd_g, (sp0, mask0, r0) of the centre point, bv0 = bv[0]
__device__ __forceinline__ NsvdFdOut nsvd_fd_combine(const float* g, float sp0, float mask0, float r0, float bv0, int D,
                                                     bool has_mask, float s_l, const nsvd_problem& prob) {
    float lap = -2.f * (float)D * g[0];
    for (int i = 0; i < D; ++i) lap += (g[1 + 2 * i] + g[2 + 2 * i]);
    const float eps2 = (float)((double)prob.eps * (double)prob.eps);
    lap = lap / eps2;
    const float spc = prob.use_importance ? fmaxf(sp0, NSVD_SQRT_P_CLAMP) : 1.f;
    lap = lap / spc;
    const float fs = g[0] / spc;
    float V;
    if (prob.potential == NSVD_POT_HYDROGEN) V = -(prob.charge_or_k / r0);
    else V = prob.charge_or_k * (r0 * r0);
    const float kinetic = -prob.scale_kinetic * lap;
    const float H = kinetic + V * fs;
    NsvdFdOut o;
    o.f = fs;
    o.Tf = prob.op_scale * (-H) + prob.op_shift * fs;
    const float w = (sp0 / spc) * prob.hard_mul_const;
    o.jac = w * mask0;
    o.dsc = has_mask ? w * bv0 * mask0 * r0 / (s_l * s_l) : 0.f;
    return o;
}

// The same central difference with the stencil points given in EVEN / ODD form (the bf16x3 forward propagates
// perturbations, pmlp_layer0_bf3.h): along direction d the head's raw output is base(x +- eps e_d) = base0 + bE[d] +- bO[d],
// and the weight w(x) = sqrt p(x) mask_l(x) at the shifted point is w0 (1 + rho_+-), rho = expm1(log w(x_+-) - log w(x0))
// with |x_+-|^2 - |x0|^2 = eps (+-2 x_d + eps) formed exactly:
//     g_+ + g_- - 2 g_0 = c w0 [ (rho_+ + rho_-)(base0 + bE) + 2 bE + (rho_+ - rho_-) bO ]
// - every term small, none the difference of two large numbers: the float32 result carries the Laplacian to ~1e-6 where
// the point-wise form (nsvd_fd_combine, the reference's own float32 arithmetic) carries it to a few per cent.
__device__ __forceinline__ NsvdFdOut nsvd_fd_evenodd(float base0, const float* bE, const float* bO, const float* xc,
                                                     int D, bool has_mask, float s_l, const nsvd_problem& prob,
                                                     float log_norm) {
    float r2 = 0.f;
    for (int d = 0; d < D; ++d) r2 = fmaf(xc[d], xc[d], r2);
    const float r0 = sqrtf(r2);
    const float c = prob.hard_mul_const;
    const float sp0 = prob.use_importance ? nsvd_sqrt_gauss_pdf(xc, D, prob.sigma, log_norm) : 1.f;
    const float mk0 = has_mask ? expf(-r0 / s_l) : 1.f;
    const float eps = prob.eps;
    const float qs = prob.use_importance ? -1.f / (4.f * prob.sigma * prob.sigma) : 0.f;  // d log sqrt p / d |x|^2
    const float e2 = eps * eps;
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
        // log w(x_+-) - log w(x0) = s +- a, split into its even part s = O(eps^2) and odd part a = O(eps) BEFORE any
        // exponential: rho_+ + rho_- is O(eps^2) while each rho is O(eps), so expm1(s + a) + expm1(s - a) loses
        // |x_d| / eps ~ 10^3 of its digits (measured: 7e-5 relative at the median of a [-50, 50] grid, i.e. the whole
        // Laplacian term); with |x_+-|^2 - |x0|^2 = e2 +- b, b = 2 x_d eps:
        //     rho_+ + rho_- = 2 [expm1(s) cosh a + (cosh a - 1)],  cosh a - 1 = 2 sinh^2(a / 2)
        //     rho_+ - rho_- = 2 exp(s) sinh a
        const float b = 2.f * xc[d] * eps;
        float sv = qs * e2, av = qs * b;
        if (has_mask) {
            // |x_+-| - |x0| = (e2 +- b) / (r_+- + r0) =: t_+-, with r_- - r_+ = -2 b / (r_+ + r_-):
            //   t_+ + t_- = [e2 (S + 2 r0) - 2 b^2 / S] / den,  t_+ - t_- = b [(S + 2 r0) - 2 e2 / S] / den,
            //   S = r_+ + r_-, den = (r_+ + r0)(r_- + r0)
            const float rp = sqrtf(fmaxf(r2 + (e2 + b), 0.f)), rm = sqrtf(fmaxf(r2 + (e2 - b), 0.f));
            const float S = rp + rm, den = (rp + r0) * (rm + r0);
            const float tsum = (e2 * (S + 2.f * r0) - 2.f * b * b / S) / den;
            const float tdif = b * ((S + 2.f * r0) - 2.f * e2 / S) / den;
            sv -= 0.5f * tsum / s_l;
            av -= 0.5f * tdif / s_l;
        }
        const float sh = sinhf(0.5f * av), chm1 = 2.f * sh * sh, es1 = expm1f(sv);
        const float ev = 2.f * (es1 * (1.f + chm1) + chm1);  // rho_+ + rho_-
        const float od = 2.f * (1.f + es1) * sinhf(av);      // rho_+ - rho_-
        acc += ev * (base0 + bE[d]) + 2.f * bE[d] + od * bO[d];
    }
    const float eps2 = (float)((double)prob.eps * (double)prob.eps);
    const float spc = prob.use_importance ? fmaxf(sp0, NSVD_SQRT_P_CLAMP) : 1.f;
    const float lap = ((c * (sp0 * mk0)) * acc / eps2) / spc;
    const float fs = (sp0 * (c * base0 * mk0)) / spc;
    float V;
    if (prob.potential == NSVD_POT_HYDROGEN) V = -(prob.charge_or_k / r0);
    else V = prob.charge_or_k * (r0 * r0);
    const float H = -prob.scale_kinetic * lap + V * fs;
    NsvdFdOut o;
    o.f = fs;
    o.Tf = prob.op_scale * (-H) + prob.op_shift * fs;
    const float w = (sp0 / spc) * c;
    o.jac = w * mk0;
    o.dsc = has_mask ? w * base0 * mk0 * r0 / (s_l * s_l) : 0.f;
    return o;
}

__device__ __forceinline__ NsvdFdOut nsvd_fd_point(const float* bv, const float* xc, int D, bool has_mask, float s_l,
                                                   const nsvd_problem& prob, float log_norm) {
    const int E = 1 + 2 * D;
    float g[2 * NSVD_FD_MAXD + 1];
    float sp0 = 1.f, mask0 = 1.f, r0 = 0.f;
    for (int e = 0; e < E; ++e) {
        const NsvdFdG o = nsvd_fd_g(e, bv[e], xc, D, has_mask, s_l, prob, log_norm);
        g[e] = o.g;
        if (e == 0) {
            sp0 = o.sp;
            mask0 = o.mk;
            r0 = o.r;
        }
    }
    return nsvd_fd_combine(g, sp0, mask0, r0, bv[0], D, has_mask, s_l, prob);
}

// Exact-Laplacian mode (laplacian_eps <= 0: VectorizedLaplacian.exact_laplacian, diff_ops.py:54-61): the model's
// value, gradient and Laplacian at x come from the forward-mode jet; here the product rule with the radial factor
// u = c * sqrt p(x) * mask_l(x), whose derivatives are closed forms:
//   sqrt p = C exp(-|x|^2 / (4 sigma^2)):  grad = -x / (2 sigma^2) sqrt p,  Lap = (-D / (2 sigma^2) + |x|^2 / (4 sigma^4)) sqrt p
//   mask   = exp(-r / s):                  grad = -(x / r) / s mask,        Lap = (1 / s^2 - (D - 1) / (r s)) mask
//   Lap(u v) = u Lap v + 2 grad u . grad v + v Lap u;  then lap / clamp(sqrt p), f = g / clamp(sqrt p), Tf as above.
__device__ __forceinline__ NsvdFdOut nsvd_fd_exact(float base, const float* dbase, float lbase, const float* xc, int D,
                                                   bool has_mask, float s_l, const nsvd_problem& prob, float log_norm) {
    float r2 = 0.f;
    for (int d = 0; d < D; ++d) r2 = fmaf(xc[d], xc[d], r2);
    const float r0 = sqrtf(r2);
    const float c = prob.hard_mul_const;
    float sp = 1.f, lsp_f = 0.f, dsp_f = 0.f;  // sqrt p, Lap sqrt p / sqrt p, and grad sqrt p = dsp_f * x * sqrt p
    if (prob.use_importance) {
        sp = nsvd_sqrt_gauss_pdf(xc, D, prob.sigma, log_norm);
        const float is2 = 1.f / (2.f * prob.sigma * prob.sigma);
        dsp_f = -is2;
        lsp_f = -(float)D * is2 + r2 * is2 * is2;
    }
    float mk = 1.f, lmk_f = 0.f, dmk_f = 0.f;  // mask, Lap mask / mask, grad mask = dmk_f * x * mask
    if (has_mask) {
        mk = expf(-r0 / s_l);
        dmk_f = -1.f / (r0 * s_l);
        lmk_f = 1.f / (s_l * s_l) - (float)(D - 1) / (r0 * s_l);
    }
    const float u = c * sp * mk;
    // grad u = u (dsp_f + dmk_f) x;  Lap u = u (lsp_f + 2 dsp_f dmk_f |x|^2 + lmk_f)
    const float gu = dsp_f + dmk_f;
    float dot = 0.f;
    for (int d = 0; d < D; ++d) dot = fmaf(xc[d], dbase[d], dot);
    const float lu = lsp_f + 2.f * dsp_f * dmk_f * r2 + lmk_f;
    const float g = u * base;
    float lap = u * (lbase + 2.f * gu * dot + base * lu);
    const float spc = prob.use_importance ? fmaxf(sp, NSVD_SQRT_P_CLAMP) : 1.f;
    lap = lap / spc;
    const float fs = g / spc;
    float V;
    if (prob.potential == NSVD_POT_HYDROGEN) V = -(prob.charge_or_k / r0);
    else V = prob.charge_or_k * (r0 * r0);
    const float H = -prob.scale_kinetic * lap + V * fs;
    NsvdFdOut o;
    o.f = fs;
    o.Tf = prob.op_scale * (-H) + prob.op_shift * fs;
    const float w = (sp / spc) * c;
    o.jac = w * mk;
    o.dsc = has_mask ? w * base * mk * r0 / (s_l * s_l) : 0.f;
    return o;
}
