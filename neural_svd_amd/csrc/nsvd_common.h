// Shared host/device helpers for libnsvd_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nsvd.h"

#define NSVD_SOFTPLUS_THRESHOLD 20.0f  // torch.nn.Softplus default (reference mlp.py:84-85)
#define NSVD_SQRT_P_CLAMP 1e-5f        // reference diff_ops.py:15
#define NSVD_LOG2E 1.4426950408889634f
#define NSVD_LN2 0.6931471805599453f

#define NSVD_CHECK_LAUNCH()                          \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return -(int)e__;     \
    } while (0)

static inline int nsvd_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t nsvd_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// softplus(z) = log(1 + e^z) with torch's threshold. Hardware exp2/log2 (v_exp_f32 / v_log_f32, ~1 ulp) plus the
// first-order log1p correction: with u = fl(1 + t), d = u - 1 (exact), ln(1 + t) = ln(u) + (t - d) / u + ..., and
// since |t - d| <= 2^-24 the factor 1/u is dropped altogether: (t - d)(1 - 1/u) <= 6e-8 t / (1 + t) <= 6e-8 ln(1 + t),
// half an ulp of the result (and exact in the limit u = 1, where the correction is all there is). 2 transcendental
// and 7 VALU instructions instead of the ~60 of ocml log1pf(expf()); max(z,0) + log1p(exp(-|z|)) never
// overflows. Max relative error 1.8e-6 at z = -29 (the float32 rounding of |z| log2 e), checked against float64.
__device__ __forceinline__ float nsvd_softplus(float z) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(z) * NSVD_LOG2E);  // e^{-|z|} in (0, 1]
    const float u = 1.0f + t;
    const float d = u - 1.0f;
    const float l = fmaf(__builtin_amdgcn_logf(u), NSVD_LN2, t - d);  // ln(1 + t)
    // max(z, 0) as ONE instruction (v_med3_f32; fmaxf costs a second v_max to quieten NaNs), and no select for torch's
    // threshold: above z = 20 (NSVD_SOFTPLUS_THRESHOLD) l = e^-z < 2.1e-9 is below half an ulp of z (>= 9.5e-7), so
    // z + l rounds to z itself - bit for bit what the select returned. Every VALU instruction here costs matrix-pipe
    // time in the fused forward (scripts/experiments/README.md): 80 of these per lane and layer.
    return __builtin_amdgcn_fmed3f(z, 0.0f, __builtin_inff()) + l;
}

// d softplus / dz = sigmoid(z)  (1 above the threshold, like torch's softplus_backward)
__device__ __forceinline__ float nsvd_sigmoid(float z) {
    const float t = __builtin_amdgcn_exp2f(-fabsf(z) * NSVD_LOG2E);  // e^{-|z|}
    const float r = __builtin_amdgcn_rcpf(1.0f + t);                 // sigma(|z|)
    const float s = z >= 0.0f ? r : t * r;
    return z > NSVD_SOFTPLUS_THRESHOLD ? 1.0f : s;
}

// softplus on a stencil pair in EVEN / ODD form: z(x +- eps e_d) = z0 + zE +- zO with zO = O(delta),
// zE = O(delta^2); returns the even and odd parts of softplus(z0 + zE +- zO) - softplus(z0) by the Taylor expansion
// around z0 to sixth order in delta (s = sigmoid z0, p = s (1 - s), w = zO^2):
//   even' = s zE + c2 (zE^2 + w) + c3 zE (zE^2 + 3 w) + c4 w (w + 6 zE^2) + 5 c5 zE w^2 + c6 w^3   + O(delta^8)
//   odd'  = zO [s + 2 c2 zE + c3 (3 zE^2 + w) + 4 c4 zE w + c5 w^2]                                 + O(delta^7)
// (the fused forward kernels carry the same expansion inline, sharing the coefficients between the directions)
#define NSVD_EO_TAYLOR_MAX 0.25f  // beyond it (truncation > ~1e-5): nsvd_softplus_evenodd_large
// softplus(z0 + d) - softplus(z0) for a LARGE d (the kernels' rare path), through the mirrored identity
// softplus(z) = z + softplus(-z): with a = |z0|,
//     z0 <= 0:  softplus(d - a) - softplus(-a),        z0 > 0:  d + [softplus(-d - a) - softplus(-a)]
// - both softplus values are taken at NON-POSITIVE-centred arguments, where softplus(-a) = log1p(e^-a) is small and
// known to full relative accuracy, so their difference carries ~1e-7 of ITS OWN size (p99.9 7e-6 over wide ranges of
// z0 and d; the plain difference softplus(z0 + d) - softplus(z0) carries 1e-7 x |z0|: 1.2e-4 on the ground state's
// eigenvalue, whose cusp at the nucleus is where the large perturbations are). No clamp, no overflow, no select: two
// softplus per value - what the rare path may cost the kernel that hosts it was measured the hard way.
__device__ __forceinline__ float nsvd_softplus_diff(float a, float sp_ma, bool pos, float d) {
    return (pos ? d : 0.f) + (nsvd_softplus((pos ? -d : d) - a) - sp_ma);
}
// the even / odd parts for a LARGE perturbation (the rare path of the kernels: wide stencils, very large weights)
__device__ __forceinline__ void nsvd_softplus_evenodd_large(float z0, float zE, float zO, float* even, float* odd) {
    const float a = fabsf(z0), sp_ma = nsvd_softplus(-a);
    const bool pos = z0 > 0.f;
    const float dp = nsvd_softplus_diff(a, sp_ma, pos, zE + zO), dm = nsvd_softplus_diff(a, sp_ma, pos, zE - zO);
    *even = 0.5f * (dp + dm);
    *odd = 0.5f * (dp - dm);
}
__device__ __forceinline__ void nsvd_softplus_evenodd(float z0, float zE, float zO, float* even, float* odd) {
    if (fmaxf(fabsf(zO), fabsf(zE)) > NSVD_EO_TAYLOR_MAX) {
        nsvd_softplus_evenodd_large(z0, zE, zO, even, odd);
        return;
    }
    const float s1 = nsvd_sigmoid(z0);
    const float sq = z0 > NSVD_SOFTPLUS_THRESHOLD ? 0.f : s1 * (1.f - s1);
    const float t12 = fmaf(-2.f, s1, 1.f);
    const float c2 = 0.5f * sq;
    const float c3 = sq * t12 * (1.f / 6.f);
    const float c4 = sq * fmaf(-6.f, sq, 1.f) * (1.f / 24.f);
    const float c5 = sq * t12 * fmaf(-12.f, sq, 1.f) * (1.f / 120.f);
    const float c6 = sq * fmaf(sq, fmaf(120.f, sq, -30.f), 1.f) * (1.f / 720.f);
    const float w = zO * zO, e2 = zE * zE;
    float ev = fmaf(c6, w, 5.f * c5 * zE);
    ev = fmaf(ev, w, c4 * fmaf(6.f, e2, w));
    ev = fmaf(ev, w, c3 * zE * fmaf(3.f, w, e2));
    ev = fmaf(c2, e2 + w, ev);
    float od = fmaf(c5, w, 4.f * c4 * zE);
    od = fmaf(od, w, c3 * fmaf(3.f, e2, w));
    od = fmaf(2.f * c2, zE, od) + s1;
    *even = fmaf(s1, zE, ev);
    *odd = zO * od;
}

// The same derivative from the ACTIVATION a = softplus(z) >= 0 (what the fused forward saves for the backward, so
// that no kernel of the backward recomputes a softplus): sigmoid(z) = 1 - e^{-a}. Below a = 1/16 the series of
// -expm1(-a) to a^4 (no cancellation; truncation a^4/120 < 1.3e-7 relative), above it 1 - exp directly (the result
// is >= 0.06, so the 6e-8 absolute rounding of 1 - x is <= 1e-6 relative).
__device__ __forceinline__ float nsvd_sigmoid_from_softplus(float a) {
    const float direct = 1.0f - __builtin_amdgcn_exp2f(-a * NSVD_LOG2E);
    float p = fmaf(a, -0.25f, 1.0f);
    p = fmaf(p * a, -1.0f / 3.0f, 1.0f);
    p = fmaf(p * a, -0.5f, 1.0f);
    return a < 0.0625f ? p * a : direct;
}

// sin and cos of one float32 argument, ~1 ulp, ~30 VALU: 3-constant Cody-Waite reduction by pi/2 with
// FMAs (exact enough for |x| < 1e5) + degree-7/8 minimax polynomials on [-pi/4, pi/4]; ocml's sincosf
// (Payne-Hanek, ~10x the cost at the 10..60 rad arguments the Fourier features see) beyond that.
__device__ __forceinline__ void nsvd_sincos(float x, float* sp, float* cp) {
    if (!(fabsf(x) < 65536.0f)) {  // also catches nan / inf
        sincosf(x, sp, cp);
        return;
    }
    const float n = rintf(x * 0.636619772367581343f);  // x * 2/pi
    float r = fmaf(n, -1.57079601287841796875f, x);     // pi/2 = c1 + c2 + c3
    r = fmaf(n, -3.1391647326017846353e-7f, r);
    r = fmaf(n, -5.3903025299577647655e-15f, r);
    const float s2 = r * r;
    float ps = fmaf(s2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(ps, s2, -1.6666654611e-1f);
    const float sn = fmaf(ps * s2, r, r);
    float pc = fmaf(s2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(pc, s2, 4.166664568298827e-2f);
    const float cs = fmaf(pc * s2, s2, fmaf(-0.5f, s2, 1.0f));
    const int q = (int)n & 3;
    const float so = (q & 1) ? cs : sn;
    const float co = (q & 1) ? sn : cs;
    *sp = (q & 2) ? -so : so;
    *cp = ((q + 1) & 2) ? -co : co;
}

__device__ __forceinline__ float nsvd_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// sqrt of the isotropic Gaussian pdf N(0, sigma^2 I) the way MultivariateNormal.log_prob().exp().sqrt()
// evaluates it (reference main_pde.py:94-100): M = sum (x_d / sigma)^2.
__device__ __forceinline__ float nsvd_sqrt_gauss_pdf(const float* xr, int D, float sigma, float log_norm) {
    float M = 0.f;
    for (int d = 0; d < D; ++d) {
        const float t = xr[d] / sigma;
        M = fmaf(t, t, M);
    }
    return sqrtf(expf(-0.5f * M + log_norm));
}

// host: -0.5 D log(2 pi) - D log(sigma)
static inline float nsvd_gauss_log_norm(int D, float sigma) {
    return (float)(-0.5 * D * 1.8378770664093453 - D * log((double)sigma));
}

// stencil point e of row b: e = 0 centre, e = 1 + 2i: +eps on axis i, e = 2 + 2i: -eps
__device__ __forceinline__ float nsvd_stencil_coord(float xc, int d, int e, float eps) {
    if (e == 0) return xc;
    const int axis = (e - 1) >> 1;
    if (axis != d) return xc;
    return ((e - 1) & 1) ? xc - eps : xc + eps;
}

// ---- counter-based normal sampler (Philox4x32-10 + Box-Muller): x[b][d] ~ N(0, sigma^2) --------------------
// Replaces the reference's host-side x = sigma * randn(B, D) + copy (examples/operator/pde/main_pde.py:92-93).
// Stateless: the value of (seed, offset, b, d) is a pure function, so every block of the feature kernel can
// regenerate the coordinates of its samples instead of reading them, and ranks that must agree (head-parallel
// sharding) agree by construction.
struct NsvdSampler {
    unsigned long long seed;    // key
    unsigned long long offset;  // call counter (one per batch)
    float sigma;
    int on;                     // 0: coordinates are read from x
    // device counter added to `offset` (nsvd_step_state::step), or null: the batch counter of a step captured in a
    // HIP graph lives on the device, `offset` is then the constant base
    const unsigned long long* offset_add;
};

__host__ __device__ __forceinline__ void nsvd_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                                            unsigned k0, unsigned k1, unsigned (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// up to 4 coordinates of sample b (D <= 4): two Box-Muller pairs from one Philox block
__device__ __forceinline__ void nsvd_sample_row(const NsvdSampler& s, int b, int D, float* xr) {
    unsigned r[4];
    const unsigned long long off = s.offset + (s.offset_add ? *s.offset_add : 0ull);
    nsvd_philox4x32_10((unsigned)b, (unsigned)off, (unsigned)(off >> 32), 0x6e737664u, (unsigned)s.seed,
                       (unsigned)(s.seed >> 32), r);
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
        if (2 * pr >= D) break;
        // (0, 1) strictly: 23 random bits + 1/2 is exact in float32 (24 bits); with 24 bits k + 1/2 rounds to even and
        // k = 2^24 - 1 gives u1 = 1, a radius of exactly 0 - and the hydrogen potential -Z / |x| of that sample is -inf
        // (it happened at step 4081 of a configs[1] run: once in 2^24 pairs)
        const float u1 = ((float)(r[2 * pr] >> 9) + 0.5f) * (1.0f / 8388608.0f);
        const float u2 = ((float)(r[2 * pr + 1] >> 9) + 0.5f) * (1.0f / 8388608.0f);
        const float rad = s.sigma * sqrtf(-2.0f * logf(u1));
        float sn, cs;
        sincosf(6.283185307179586f * u2, &sn, &cs);
        xr[2 * pr] = rad * cs;
        if (2 * pr + 1 < D) xr[2 * pr + 1] = rad * sn;
    }
}
