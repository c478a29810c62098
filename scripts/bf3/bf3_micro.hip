// dev microbenchmark: layer-0 chunk loop with 3-way bf16-split operands (6 products) on v_mfma_f32_32x32x16_bf16:
// LDS fragment reads + MFMAs only (operands resident in LDS), per-chunk cycles vs the fp32 MFMA loop's 5120.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ROWB = 80;          // bytes per row per plane: 32 bf16 + 16 B pad
constexpr int NA = 128, NB = 160; // W rows, sample columns
template <int NV, int BAR = 0, int NW = 0>
__global__ void __launch_bounds__(256, 1) k(int iters, float* out, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* As = lds;                      // [3][128][80]
    char* Bs = lds + 3 * NA * ROWB;      // [3][160][80]
    for (int i = threadIdx.x; i < (3 * NA * ROWB + 3 * NB * ROWB) / 4; i += 256) ((unsigned*)lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, li = lane & 31, hi = lane >> 5;
    f32x16 acc[5];
    for (int e = 0; e < 5; ++e) for (int r = 0; r < 16; ++r) acc[e][r] = 0.f;
    float dummy[8];
    for (int i = 0; i < 8; ++i) dummy[i] = 1.0f + threadIdx.x * 1e-3f + i;
    const unsigned long long t0 = __builtin_readcyclecounter();
    char* scratch = lds + 3 * NA * ROWB + 3 * NB * ROWB;  // write target (never read)
    const uint4 wv = make_uint4(threadIdx.x, 1, 2, 3);
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");  // the fragment reads of every chunk are real
        if (BAR) __syncthreads();
#pragma unroll
        for (int v = 0; v < NW; ++v) *(uint4*)(scratch + (v * 256 + threadIdx.x) * 16) = wv;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[3], b[5][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) a[p] = *(const bf16x8*)(As + (p * NA + 32 * w + li) * ROWB + ks * 32 + hi * 16);
#pragma unroll
            for (int e = 0; e < 5; ++e)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[e][p] = *(const bf16x8*)(Bs + (p * NB + 32 * e + li) * ROWB + ks * 32 + hi * 16);
            const int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[t]], b[e][TB[t]], acc[e], 0, 0, 0);
                    // NV independent VALU instructions per MFMA (8 separate dependency chains)
#pragma unroll
                    for (int v = 0; v < NV; ++v) dummy[v & 7] = __builtin_fmaf(dummy[v & 7], 1.0000001f, 1e-7f);
#pragma unroll
                    for (int v = 0; v < (NV > 0 ? 1 : 0); ++v) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
                    }
                }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int e = 0; e < 5; ++e) for (int r = 0; r < 16; ++r) s += acc[e][r];
    for (int i = 0; i < 8; ++i) s += dummy[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int iters = 2000, nb = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nb * 256 * 4); hipMalloc(&cyc, nb * 8);
    const size_t lds = 3 * NA * ROWB + 3 * NB * ROWB + 14 * 4096;
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<0, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<0, 0, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)k<0, 1, 14>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 6; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        if (rep == 0) k<0><<<nb, 256, lds>>>(iters, out, cyc);
        if (rep == 1) k<3><<<nb, 256, lds>>>(iters, out, cyc);
        if (rep == 2) k<6><<<nb, 256, lds>>>(iters, out, cyc);
        if (rep == 3) k<0, 1, 0><<<nb, 256, lds>>>(iters, out, cyc);
        if (rep == 4) k<0, 0, 14><<<nb, 256, lds>>>(iters, out, cyc);
        if (rep == 5) k<0, 1, 14><<<nb, 256, lds>>>(iters, out, cyc);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(nb); hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
        double flops = (double)nb * 4 * iters * 60 * 32.0 * 32 * 16 * 2;
        const char* names[6] = {"NV=0", "NV=3", "NV=6", "NV=0 + barrier per chunk", "NV=0 + 14 ds_write_b128 per thread per chunk", "NV=0 + barrier + writes"};
        printf("%s: ", names[rep]);
        printf("NV=%d VALU per MFMA: %.3f ms, %.0f cycles/chunk (fp32 loop: 5120 ideal, ~5400 measured), bf16 MFMA rate %.0f TFLOP/s, fp32-equivalent %.0f TFLOP/s\n",
               rep < 3 ? rep * 3 : 0, ms, (double)h[0] / iters, flops / ms * 1e-9, flops / 6 / ms * 1e-9);
    }
    return 0;
}
