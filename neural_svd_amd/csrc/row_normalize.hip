// normalize(z, r_up, 'l2_ball' | 'l2_sphere') of the CDK towers (reference examples/models/siam.py:170-183):
//   l2_ball:   rows with ||z|| <  r_up pass through, the others become r_up * z / max(||z||, 1e-12)
//   l2_sphere: every row becomes                                   r_up * z / max(||z||, 1e-12)
// and its backward (the comparison mask is a constant, as in the reference's mask * z + (1 - mask) * ... form):
//   pass-through rows: dz = dout;  scaled rows: dz = r_up * (dout / n - z (z . dout) / n^3), n = max(||z||, 1e-12)
//   (rows below the 1e-12 clamp have a constant divisor: dz = r_up * dout / n).
// HBM-bound elementwise work: one wave per row, 16-byte accesses when L % 4 == 0, two passes over a row that is
// L2-resident after the first.
#include "nsvd_kernels.h"

namespace {

constexpr float NRM_EPS = 1e-12f;  // torch.nn.functional.normalize's default eps

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool BWD>
__global__ void __launch_bounds__(256) row_normalize_kernel(const float* __restrict__ z, const float* __restrict__ dout,
                                                            int B, int L, float r, int sphere, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const float* zr = z + (size_t)row * L;
    const float* dr = BWD ? dout + (size_t)row * L : nullptr;
    float* orow = out + (size_t)row * L;
    const bool vec = (L & 3) == 0;
    float ss = 0.f, zd = 0.f;
    if (vec) {
        for (int i = lane * 4; i < L; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(zr + i);
            ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            if (BWD) {
                const float4 d = *reinterpret_cast<const float4*>(dr + i);
                zd += (v.x * d.x + v.y * d.y) + (v.z * d.z + v.w * d.w);
            }
        }
    } else {
        for (int i = lane; i < L; i += 64) {
            ss = fmaf(zr[i], zr[i], ss);
            if (BWD) zd = fmaf(zr[i], dr[i], zd);
        }
    }
    ss = wave_sum(ss);
    if (BWD) zd = wave_sum(zd);
    const float nrm = sqrtf(ss);
    const bool pass = !sphere && nrm < r;
    const float n = fmaxf(nrm, NRM_EPS);
    // forward: out = a z;  backward: dz = a dout - b z
    const float a = pass ? 1.f : r / n;
    const float b = (BWD && !pass && nrm >= NRM_EPS) ? r * zd / (n * n * n) : 0.f;
    if (vec) {
        for (int i = lane * 4; i < L; i += 256) {
            const float4 v = *reinterpret_cast<const float4*>(zr + i);
            float4 o;
            if (BWD) {
                const float4 d = *reinterpret_cast<const float4*>(dr + i);
                o = make_float4(a * d.x - b * v.x, a * d.y - b * v.y, a * d.z - b * v.z, a * d.w - b * v.w);
            } else {
                o = make_float4(a * v.x, a * v.y, a * v.z, a * v.w);
            }
            *reinterpret_cast<float4*>(orow + i) = o;
        }
    } else {
        for (int i = lane; i < L; i += 64) orow[i] = BWD ? a * dr[i] - b * zr[i] : a * zr[i];
    }
}

int check(const float* z, int B, int L, float r, int mode, const float* out) {
    if (!z || !out || B <= 0 || L <= 0 || !(r > 0.f) || (mode != 0 && mode != 1)) return NSVD_EINVAL;
    return 0;
}

}  // namespace

extern "C" int nsvd_row_normalize_forward(const float* z, int B, int L, float r_up, int mode, float* out,
                                          void* stream) {
    if (int rc = check(z, B, L, r_up, mode, out)) return rc;
    hipLaunchKernelGGL(row_normalize_kernel<false>, dim3(nsvd_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, z, nullptr,
                       B, L, r_up, mode, out);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_row_normalize_backward(const float* z, const float* dout, int B, int L, float r_up, int mode,
                                           float* dz, void* stream) {
    if (int rc = check(z, B, L, r_up, mode, dz)) return rc;
    if (!dout) return NSVD_EINVAL;
    hipLaunchKernelGGL(row_normalize_kernel<true>, dim3(nsvd_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, z, dout, B,
                       L, r_up, mode, dz);
    NSVD_CHECK_LAUNCH();
    return 0;
}
