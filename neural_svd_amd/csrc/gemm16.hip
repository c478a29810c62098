// nsvd_gemm_bf16 (include/nsvd.h): the bf16-MFMA contraction of the mixed-precision towers as an entry point of its
// own (the tile kernel: gemm16.h), plus the float32 -> bfloat16 cast the towers' operands go through.
#include <string.h>
#include "nsvd_kernels.h"
#include "gemm16.h"

namespace {

// float32 -> bfloat16 (round to nearest even: v_cvt_pk_bf16_f32), 8 values per thread and pass
__global__ void __launch_bounds__(256) to_bf16_kernel(const float4* __restrict__ in, uint4* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const float4 a = in[2 * i], b = in[2 * i + 1];
        uint4 o;
        o.x = nsvd_g16::pack_bf16(a.x, a.y);
        o.y = nsvd_g16::pack_bf16(a.z, a.w);
        o.z = nsvd_g16::pack_bf16(b.x, b.y);
        o.w = nsvd_g16::pack_bf16(b.z, b.w);
        out[i] = o;
    }
}

}  // namespace

int nsvd_to_bf16_launch(const float* in, void* out, size_t n, hipStream_t s) {
    if (!in || !out || (n % 8) || (((uintptr_t)in | (uintptr_t)out) & 15)) return NSVD_EINVAL;
    const size_t n8 = n / 8;
    if (n8 == 0) return 0;
    size_t blocks = (n8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(to_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float4*)in, (uint4*)out, n8);
    NSVD_CHECK_LAUNCH();
    return 0;
}

extern "C" int nsvd_to_bf16(const float* in, void* out, size_t n, void* stream) {
    return nsvd_to_bf16_launch(in, out, n, (hipStream_t)stream);
}

static unsigned long long* g_dbg_stamps = nullptr;

extern "C" int nsvd_gemm_bf16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda,
                              long ldb, long ldc, int a_kstrided, int b_kstrided, int out_bf16, int slices,
                              long slice_stride, float* sumsq, void* stream) {
    nsvd_g16::Args a;
    memset(&a, 0, sizeof(a));
    a.p[0].A = A; a.p[0].B = B; a.p[0].C = C; a.p[0].bias = bias; a.p[0].sumsq = sumsq;
    a.nprob = 1; a.K = K; a.S = slices; a.slice_stride = slice_stride;
    nsvd_g16::set_uniform(a, M, N, lda, ldb, ldc);
    a.stamps = g_dbg_stamps;  // (null unless nsvd_debug_g16_stamps armed it)
    return nsvd_g16::launch(a, a_kstrided != 0, b_kstrided != 0, out_bf16 != 0, (hipStream_t)stream);
}

// developer diagnostic (not in include/nsvd.h): the first call arms the stamps of nsvd_gemm_bf16's launches (cycles of
// block 0 / wave 0: prologue, K-loop halves and barrier per step, epilogue, K steps), later calls read the last launch's
extern "C" int nsvd_debug_g16_stamps(unsigned long long* host) {
    if (!g_dbg_stamps) {
        hipError_t e = hipMalloc((void**)&g_dbg_stamps, 8 * sizeof(unsigned long long));
        if (e != hipSuccess) return -(int)e;
        return -(int)hipMemset(g_dbg_stamps, 0, 8 * sizeof(unsigned long long));
    }
    return -(int)hipMemcpy(host, g_dbg_stamps, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

