// dev microbenchmark: do LDS-DMA loads (global_load_lds_dwordx4) and ordinary register loads retire in issue order with
// respect to each other, i.e. may "s_waitcnt vmcnt(N)" be used to wait for an older DMA while N younger register loads
// (or an older register load while N younger DMAs) stay in flight?
//   test A: DMA from a cold (never touched) line, then a register load from a hot line, s_waitcnt vmcnt(1), read LDS.
//   test B: register load from a cold line, then a DMA from a hot line, s_waitcnt vmcnt(1), use the register.
// A mismatch count of 0 over many trials = in order (for this access pattern).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void __launch_bounds__(64) k(const unsigned* cold, const unsigned* hot, int trials, size_t stride, unsigned* bad, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x;
    unsigned nbad = 0;
    const unsigned ldsaddr = (unsigned)(size_t)lds;
    volatile unsigned h0 = hot[lane];  // make the hot line hot
    (void)h0;
    for (int t = 0; t < trials; ++t) {
        const unsigned* c = cold + ((size_t)blockIdx.x * trials + t) * stride + lane * 4;  // 16 B per lane, fresh lines
        const unsigned* h = hot + lane;
        lds[lane * 4] = 0xdeadbeefu;
        __syncthreads();
        unsigned r = 0;
        if (mode == 0) {
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                         "global_load_dword %0, %3, off\n\t"
                         "s_waitcnt vmcnt(1)"
                         : "=v"(r) : "v"(c), "s"(ldsaddr), "v"(h) : "memory", "m0");
            const unsigned got = lds[lane * 4];          // must be the DMA's data if loads retire in order
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (got != c[0]) ++nbad;
        } else {
            unsigned r2;
            asm volatile("global_load_dword %0, %2, off\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
                         "s_waitcnt vmcnt(1)\n\t"
                         "v_mov_b32 %1, %0"
                         : "=&v"(r), "=&v"(r2) : "v"(c), "s"(ldsaddr), "v"(h) : "memory", "m0");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (r2 != r) ++nbad;                          // r2 = the register as seen right after vmcnt(1)
        }
        __syncthreads();
    }
    atomicAdd(bad, nbad);
}
int main() {
    const int nb = 1024, trials = 200;
    const size_t stride = 4096;  // dwords between trials: fresh lines every time
    unsigned *cold, *hot, *bad;
    const size_t n = (size_t)nb * trials * stride + 1024;
    hipMalloc(&cold, n * 4); hipMalloc(&hot, 4096); hipMalloc(&bad, 4);
    std::vector<unsigned> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 2654435761u + 12345u);
    hipMemcpy(cold, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(hot, 1, 4096);
    for (int mode = 0; mode < 2; ++mode) {
        hipMemset(bad, 0, 4);
        k<<<nb, 64, 1024>>>(cold, hot, trials, stride, bad, mode);
        hipDeviceSynchronize();
        unsigned b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("%s: %u mismatching lanes of %d\n", mode == 0 ? "A: older DMA (cold), younger register load (hot), vmcnt(1)"
                                                            : "B: older register load (cold), younger DMA (hot), vmcnt(1)",
               b, nb * trials * 64);
    }
    return 0;
}
