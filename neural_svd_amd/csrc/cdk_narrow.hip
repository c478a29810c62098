// The NARROW end of a CDK training step (B x d2, d2 = 512 at BASELINE configs[4]) for both towers at once:
//   forward   Y2 = b2 + sum of the split-K partial products of A1 W2^T     (examples/models/mlp.py:129-164: Linear2)
//             Z  = BatchNorm1d(Y2) in training mode, running statistics     (the second BatchNorm of get_mlp)
//             E  = normalize(Z, sqrt(mu), 'l2_ball' | 'l2_sphere')          (examples/models/siam.py:170-183)
//   backward  dZ = normalize'(dE), dgamma2, dbeta2, db2, dY2 = BatchNorm'(dZ)
// As stage calls (tower.hip strips of 4 columns + row_normalize.hip, per tower) this is 2 MB of data per tower moved by
// six latency-bound launches: 47 + 38 us of a 400 us step. Here: three launches each way for BOTH towers -
//   (1) rows in blocks of 8 (one workgroup per block and tower): the row-wise work, and per-block column partials
//       (count, mean, M2 - merged by Chan's formula, never sum-of-squares minus square-of-sum; the backward's plain sums)
//   (2) one small launch that merges the partials per column (fixed order) into the statistics
//   (3) the elementwise pass that needs them.
// Used by nsvd_cdk_step in mixed precision (cdk_step.hip); the stage entry points keep their own kernels.
#include "nsvd_kernels.h"

namespace {

constexpr int RB = 8;         // rows per workgroup of the row-block kernels (B / 8 workgroups per tower: 256 at B = 1024, nt = 2)
constexpr int FG = 16;        // block groups of the finish kernels (16 columns x 16 groups per workgroup: at B = 1024 a
                              // thread has 8 blocks to merge - ONE batch of loads, one memory latency)
constexpr int NTH = 256;
constexpr int FC = NTH / FG;  // columns per workgroup of the finish kernels
constexpr float NRM_EPS = 1e-12f;  // torch.nn.functional.normalize's default eps

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------- forward (1)
// Y2 = bias + sum of slices (slice order); per block and column: mean over the block's RB rows and M2 about it.
// thread t: column quad t % (N / 4) ... N = 4 * 128 at most per pass; rows (t / (N / 4)) * RPT .. + RPT
__global__ void __launch_bounds__(NTH) narrow_sum_stats_kernel(NsvdNarrowFwd a) {
    __shared__ float red[2][NTH * 4];
    const int t = blockIdx.y, rb = blockIdx.x, tid = threadIdx.x;
    const int Q = a.N / 4;             // column quads (<= 256)
    const int groups = NTH / Q;        // row groups of the workgroup (1, 2 or 4 ..)
    const int rpt = RB / groups;       // rows per thread
    const int cq = tid % Q, rg = tid / Q;
    const float* part = a.Y2p[t];
    const float4 bv = a.bias[t] ? *reinterpret_cast<const float4*>(a.bias[t] + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v[RB];
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        if (k < rpt) {
            const size_t off = (size_t)(rb * RB + rg * rpt + k) * a.N + 4 * cq;
            float4 acc = bv;
            for (int s0 = 0; s0 < a.S; s0 += 8) {
                float4 u[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    u[j] = s0 + j < a.S ? *reinterpret_cast<const float4*>(part + (size_t)(s0 + j) * a.slice_stride + off)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc.x += u[j].x; acc.y += u[j].y; acc.z += u[j].z; acc.w += u[j].w;
                }
            }
            v[k] = acc;
            *reinterpret_cast<float4*>(a.Y2[t] + off) = acc;
            s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        }
    }
    // block mean per column: the row groups' sums through LDS, added in group order
    float* r0 = red[0];
    r0[4 * tid + 0] = s1.x; r0[4 * tid + 1] = s1.y; r0[4 * tid + 2] = s1.z; r0[4 * tid + 3] = s1.w;
    __syncthreads();
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < groups; ++g) {
        const float* p = r0 + 4 * (g * Q + cq);
        mu.x += p[0]; mu.y += p[1]; mu.z += p[2]; mu.w += p[3];
    }
    const float rn = 1.0f / (float)RB;
    mu.x *= rn; mu.y *= rn; mu.z *= rn; mu.w *= rn;
    float4 s2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RB; ++k) {
        if (k < rpt) {
            const float dx = v[k].x - mu.x, dy = v[k].y - mu.y, dz = v[k].z - mu.z, dw = v[k].w - mu.w;
            s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
        }
    }
    float* r1 = red[1];
    r1[4 * tid + 0] = s2.x; r1[4 * tid + 1] = s2.y; r1[4 * tid + 2] = s2.z; r1[4 * tid + 3] = s2.w;
    __syncthreads();
    if (rg == 0) {
        float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int g = 0; g < groups; ++g) {
            const float* p = r1 + 4 * (g * Q + cq);
            m2.x += p[0]; m2.y += p[1]; m2.z += p[2]; m2.w += p[3];
        }
        // partials: [tower][block][2][N]
        float* out = a.part + ((size_t)(t * gridDim.x + rb) * 2) * a.N + 4 * cq;
        *reinterpret_cast<float4*>(out) = mu;
        *reinterpret_cast<float4*>(out + a.N) = m2;
    }
}

// ---------------------------------------------------------------------------------------------------- forward (2)
// per column: merge the blocks' (mean, M2) (Chan et al.) in a fixed order - FG groups of consecutive blocks, each merged
// in block order by its own thread (the group's partials are loaded before any is used: one memory latency, not one per
// block), then the four groups in order - then mean / invstd / running statistics. Workgroup = 64 columns x 4 groups.
__global__ void __launch_bounds__(NTH) narrow_stats_finish_kernel(NsvdNarrowFwd a, int nblocks) {
    __shared__ float gm[FG][FC], gq[FG][FC], gn[FG][FC];
    const int t = blockIdx.y;
    const int cl = threadIdx.x % FC, grp = threadIdx.x / FC;
    const int c = blockIdx.x * FC + cl;
    const int per = (nblocks + FG - 1) / FG;
    const int b0 = grp * per, b1 = min(nblocks, b0 + per);
    float mean = 0.f, m2 = 0.f, n = 0.f;
    if (c < a.N) {
        const float* p = a.part + (size_t)t * nblocks * 2 * a.N + c;
        for (int bb = b0; bb < b1; bb += 16) {
            float mb[16], qb[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = bb + j < b1;
                mb[j] = ok ? p[(size_t)(bb + j) * 2 * a.N] : 0.f;
                qb[j] = ok ? p[(size_t)(bb + j) * 2 * a.N + a.N] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (bb + j < b1) {
                    const float nb = (float)RB, nn = n + nb;
                    const float d = mb[j] - mean;
                    mean += d * (nb / nn);
                    m2 += qb[j] + d * d * (n * nb / nn);
                    n = nn;
                }
            }
        }
    }
    gm[grp][cl] = mean; gq[grp][cl] = m2; gn[grp][cl] = n;
    __syncthreads();
    if (grp != 0 || c >= a.N) return;
    for (int g = 1; g < FG; ++g) {
        const float nb = gn[g][cl];
        if (nb > 0.f) {
            const float nn = n + nb, d = gm[g][cl] - mean;
            mean += d * (nb / nn);
            m2 += gq[g][cl] + d * d * (n * nb / nn);
            n = nn;
        }
    }
    const float inv = 1.0f / sqrtf(m2 / (float)a.B + a.eps);
    a.mean[t][c] = mean;
    a.invstd[t][c] = inv;
    if (a.running_mean[t]) {
        a.running_mean[t][c] = (1.f - a.momentum) * a.running_mean[t][c] + a.momentum * mean;
        a.running_var[t][c] = (1.f - a.momentum) * a.running_var[t][c] + a.momentum * (m2 / (float)(a.B - 1));
    }
}

// ---------------------------------------------------------------------------------------------------- forward (3)
// one wave per row: z = BN(Y2), e = normalize(z)
__global__ void __launch_bounds__(NTH) narrow_bn_normalize_kernel(NsvdNarrowFwd a) {
    const int t = blockIdx.y, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.B) return;
    const float* y = a.Y2[t] + (size_t)row * a.N;
    float* z = a.z[t] + (size_t)row * a.N;
    float* e = a.e[t] + (size_t)row * a.N;
    float ss = 0.f;
    for (int i = lane * 4; i < a.N; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(y + i);
        const float4 mu = *reinterpret_cast<const float4*>(a.mean[t] + i), iv = *reinterpret_cast<const float4*>(a.invstd[t] + i);
        const float4 ga = *reinterpret_cast<const float4*>(a.gamma[t] + i), be = *reinterpret_cast<const float4*>(a.beta[t] + i);
        float4 o;
        o.x = fmaf((v.x - mu.x) * iv.x, ga.x, be.x); o.y = fmaf((v.y - mu.y) * iv.y, ga.y, be.y);
        o.z = fmaf((v.z - mu.z) * iv.z, ga.z, be.z); o.w = fmaf((v.w - mu.w) * iv.w, ga.w, be.w);
        *reinterpret_cast<float4*>(z + i) = o;
        ss += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
    }
    ss = wsum(ss);
    const float nrm = sqrtf(ss);
    const bool pass = !a.sphere && nrm < a.r_up;
    const float sc = pass ? 1.f : a.r_up / fmaxf(nrm, NRM_EPS);
    for (int i = lane * 4; i < a.N; i += 256) {  // (this lane's own stores: visible to it)
        const float4 o = *reinterpret_cast<const float4*>(z + i);
        *reinterpret_cast<float4*>(e + i) = make_float4(sc * o.x, sc * o.y, sc * o.z, sc * o.w);
    }
}

// ---------------------------------------------------------------------------------------------------- backward (1)
// one workgroup = RB rows (4 waves x RB / 4 rows): dz = normalize'(ge) row by row; per block and column the sums of dz,
// dz * yhat and yhat over the block's rows (wave order, then row order within the wave: fixed)
__global__ void __launch_bounds__(NTH) narrow_bwd_rows_kernel(NsvdNarrowBwd a) {
    __shared__ float red[4][3][1024];  // [wave][sum kind][column]  (N <= 1024)
    const int t = blockIdx.y, rb = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ncol = a.N;
    const float lscale = a.loss_scale ? *a.loss_scale : 1.f;  // (a power of two: exact)
    float s0[4][4], s1[4][4], s2[4][4];  // [pass over the columns][component]: this lane's columns lane * 4 + 256 p
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) s0[p][c] = s1[p][c] = s2[p][c] = 0.f;
    for (int rr = 0; rr < RB / 4; ++rr) {
        const int row = rb * RB + (RB / 4) * w + rr;
        const float* z = a.z[t] + (size_t)row * ncol;
        const float* d = a.ge[t] + (size_t)row * ncol;
        float ss = 0.f, zd = 0.f;
        float4 zv[4], dv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = lane * 4 + 256 * p;
            if (i < ncol) {
                zv[p] = *reinterpret_cast<const float4*>(z + i);
                dv[p] = *reinterpret_cast<const float4*>(d + i);
                dv[p].x *= lscale; dv[p].y *= lscale; dv[p].z *= lscale; dv[p].w *= lscale;
                ss += (zv[p].x * zv[p].x + zv[p].y * zv[p].y) + (zv[p].z * zv[p].z + zv[p].w * zv[p].w);
                zd += (zv[p].x * dv[p].x + zv[p].y * dv[p].y) + (zv[p].z * dv[p].z + zv[p].w * dv[p].w);
            }
        }
        ss = wsum(ss);
        zd = wsum(zd);
        const float nrm = sqrtf(ss);
        const bool pass = !a.sphere && nrm < a.r_up;
        const float n = fmaxf(nrm, NRM_EPS);
        const float ca = pass ? 1.f : a.r_up / n;
        const float cb = (!pass && nrm >= NRM_EPS) ? a.r_up * zd / (n * n * n) : 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = lane * 4 + 256 * p;
            if (i < ncol) {
                const float4 y = *reinterpret_cast<const float4*>(a.Y2[t] + (size_t)row * ncol + i);
                const float4 mu = *reinterpret_cast<const float4*>(a.mean[t] + i);
                const float4 iv = *reinterpret_cast<const float4*>(a.invstd[t] + i);
                float4 g;
                g.x = ca * dv[p].x - cb * zv[p].x; g.y = ca * dv[p].y - cb * zv[p].y;
                g.z = ca * dv[p].z - cb * zv[p].z; g.w = ca * dv[p].w - cb * zv[p].w;
                *reinterpret_cast<float4*>(a.dz[t] + (size_t)row * ncol + i) = g;
                const float yx = (y.x - mu.x) * iv.x, yy = (y.y - mu.y) * iv.y, yz = (y.z - mu.z) * iv.z, yw = (y.w - mu.w) * iv.w;
                s0[p][0] += g.x; s0[p][1] += g.y; s0[p][2] += g.z; s0[p][3] += g.w;
                s1[p][0] = fmaf(g.x, yx, s1[p][0]); s1[p][1] = fmaf(g.y, yy, s1[p][1]);
                s1[p][2] = fmaf(g.z, yz, s1[p][2]); s1[p][3] = fmaf(g.w, yw, s1[p][3]);
                s2[p][0] += yx; s2[p][1] += yy; s2[p][2] += yz; s2[p][3] += yw;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = lane * 4 + 256 * p;
        if (i < ncol)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                red[w][0][i + c] = s0[p][c];
                red[w][1][i + c] = s1[p][c];
                red[w][2][i + c] = s2[p][c];
            }
    }
    __syncthreads();
    // partials: [tower][block][3][N]
    float* out = a.part + ((size_t)(t * gridDim.x + rb) * 3) * ncol;
    for (int i = threadIdx.x; i < 3 * ncol; i += NTH) {
        const int k = i / ncol, c = i - k * ncol;
        out[i] = (red[0][k][c] + red[1][k][c]) + (red[2][k][c] + red[3][k][c]);
    }
}

// ---------------------------------------------------------------------------------------------------- backward (2)
// per column: dbeta = sum dz, dgamma = sum dz yhat; the bias gradient of the Linear in front of the BatchNorm is the
// column sum of dY = gamma invstd (dz - mean(dz) - yhat mean(dz yhat)) = gamma invstd (- mean(dz yhat) sum(yhat)):
// rounding noise around zero, as torch's own (sum(yhat) = 0 in exact arithmetic)
__global__ void __launch_bounds__(NTH) narrow_bwd_finish_kernel(NsvdNarrowBwd a, int nblocks) {
    __shared__ float gs[FG][3][FC];
    __shared__ float sqp[FC];
    const int t = blockIdx.y;
    const int cl = threadIdx.x % FC, grp = threadIdx.x / FC;
    const int c = blockIdx.x * FC + cl;
    const int per = (nblocks + FG - 1) / FG;
    const int b0 = grp * per, b1 = min(nblocks, b0 + per);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (c < a.N) {  // (FG groups of consecutive blocks, each summed in block order; the groups then in order)
        const float* p = a.part + (size_t)t * nblocks * 3 * a.N + c;
        for (int bb = b0; bb < b1; bb += 16) {
            float u0[16], u1[16], u2[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = bb + j < b1;
                u0[j] = ok ? p[(size_t)(bb + j) * 3 * a.N] : 0.f;
                u1[j] = ok ? p[(size_t)(bb + j) * 3 * a.N + a.N] : 0.f;
                u2[j] = ok ? p[(size_t)(bb + j) * 3 * a.N + 2 * a.N] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                s0 += u0[j]; s1 += u1[j]; s2 += u2[j];
            }
        }
    }
    gs[grp][0][cl] = s0; gs[grp][1][cl] = s1; gs[grp][2][cl] = s2;
    __syncthreads();
    if (grp != 0 || c >= a.N) return;
    for (int g = 1; g < FG; ++g) {
        s0 += gs[g][0][cl]; s1 += gs[g][1][cl]; s2 += gs[g][2][cl];
    }
    a.dbeta[t][c] = s0;
    a.dgamma[t][c] = s1;
    const float rB = 1.0f / (float)a.B;
    a.m1[t][c] = s0 * rB;
    a.m2[t][c] = s1 * rB;
    const float db = a.gamma[t][c] * a.invstd[t][c] * ((s0 - (float)a.B * (s0 * rB)) - (s1 * rB) * s2);
    a.dbias[t][c] = db;
    if (a.sumsq[t]) {  // this workgroup's FC columns (threads 0 .. FC - 1: group 0), added in column order
        sqp[cl] = fmaf(s0, s0, fmaf(s1, s1, db * db));
        __builtin_amdgcn_wave_barrier();  // (FC <= 64: one wave wrote sqp)
        if (cl == 0) {
            float q = 0.f;
            for (int i = 0; i < FC; ++i) q += sqp[i];
            a.sumsq[t][blockIdx.x] = q;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- backward (3)
// dY2 = gamma invstd (dz - m1 - yhat m2), stored as bfloat16 (mixed precision) or float32
__global__ void __launch_bounds__(NTH) narrow_bwd_apply_kernel(NsvdNarrowBwd a) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int t = blockIdx.y;
    const size_t q = (size_t)blockIdx.x * NTH + threadIdx.x;  // float4 index
    const size_t n4 = (size_t)a.B * a.N / 4;
    if (q >= n4) return;
    const int c = (int)((q * 4) % (size_t)a.N);
    const float4 g = *reinterpret_cast<const float4*>(a.dz[t] + 4 * q);
    const float4 y = *reinterpret_cast<const float4*>(a.Y2[t] + 4 * q);
    const float4 mu = *reinterpret_cast<const float4*>(a.mean[t] + c), iv = *reinterpret_cast<const float4*>(a.invstd[t] + c);
    const float4 ga = *reinterpret_cast<const float4*>(a.gamma[t] + c);
    const float4 m1 = *reinterpret_cast<const float4*>(a.m1[t] + c), m2 = *reinterpret_cast<const float4*>(a.m2[t] + c);
    float4 o;
    o.x = ga.x * iv.x * (g.x - m1.x - (y.x - mu.x) * iv.x * m2.x);
    o.y = ga.y * iv.y * (g.y - m1.y - (y.y - mu.y) * iv.y * m2.y);
    o.z = ga.z * iv.z * (g.z - m1.z - (y.z - mu.z) * iv.z * m2.z);
    o.w = ga.w * iv.w * (g.w - m1.w - (y.w - mu.w) * iv.w * m2.w);
    if (a.dy_bf16 == 2) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        reinterpret_cast<uint2*>(a.dY[t])[q] =
            make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((f2){o.x, o.y}, h2)),
                       __builtin_bit_cast(unsigned, __builtin_convertvector((f2){o.z, o.w}, h2)));
    } else if (a.dy_bf16) {
        reinterpret_cast<uint2*>(a.dY[t])[q] =
            make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector((f2){o.x, o.y}, bf2)),
                       __builtin_bit_cast(unsigned, __builtin_convertvector((f2){o.z, o.w}, bf2)));
    } else {
        reinterpret_cast<float4*>(a.dY[t])[q] = o;
    }
}

bool narrow_ok(int nt, int B, int N) {
    return nt >= 1 && nt <= 2 && B > 0 && B % RB == 0 && N > 0 && N % 64 == 0 && N <= 1024 && NTH % (N / 4) == 0 &&
           RB % (NTH / (N / 4)) == 0;
}

}  // namespace

size_t nsvd_narrow_scratch_floats(int nt, int B, int N) { return (size_t)nt * (B / RB) * 3 * N + (size_t)nt * 2 * N; }

bool nsvd_narrow_supported(int nt, int B, int N) { return narrow_ok(nt, B, N); }

int nsvd_narrow_sumsq_count(int N) { return nsvd_cdiv(N, FC); }

int nsvd_narrow_forward(const NsvdNarrowFwd& a, hipStream_t s) {
    if (!narrow_ok(a.nt, a.B, a.N) || a.S < 1 || !a.part) return NSVD_EINVAL;
    const int nblocks = a.B / RB;
    hipLaunchKernelGGL(narrow_sum_stats_kernel, dim3(nblocks, a.nt), dim3(NTH), 0, s, a);
    NSVD_CHECK_LAUNCH();
    hipLaunchKernelGGL(narrow_stats_finish_kernel, dim3(nsvd_cdiv(a.N, FC), a.nt), dim3(NTH), 0, s, a, nblocks);
    NSVD_CHECK_LAUNCH();
    hipLaunchKernelGGL(narrow_bn_normalize_kernel, dim3(nsvd_cdiv(a.B, 4), a.nt), dim3(NTH), 0, s, a);
    NSVD_CHECK_LAUNCH();
    return 0;
}

int nsvd_narrow_backward(const NsvdNarrowBwd& a0, hipStream_t s) {
    NsvdNarrowBwd a = a0;
    if (!narrow_ok(a.nt, a.B, a.N) || !a.part) return NSVD_EINVAL;
    const int nblocks = a.B / RB;
    for (int t = 0; t < a.nt; ++t) {  // the per-column means live behind the partials
        a.m1[t] = a.part + (size_t)a.nt * nblocks * 3 * a.N + (size_t)(2 * t) * a.N;
        a.m2[t] = a.m1[t] + a.N;
    }
    hipLaunchKernelGGL(narrow_bwd_rows_kernel, dim3(nblocks, a.nt), dim3(NTH), 0, s, a);
    NSVD_CHECK_LAUNCH();
    hipLaunchKernelGGL(narrow_bwd_finish_kernel, dim3(nsvd_cdiv(a.N, FC), a.nt), dim3(NTH), 0, s, a, nblocks);
    NSVD_CHECK_LAUNCH();
    const size_t n4 = (size_t)a.B * a.N / 4;
    hipLaunchKernelGGL(narrow_bwd_apply_kernel, dim3((unsigned)((n4 + NTH - 1) / NTH), a.nt), dim3(NTH), 0, s, a);
    NSVD_CHECK_LAUNCH();
    return 0;
}
