// Weight gradients with the dW_0 optimiser epilogue off the critical path: one workgroup of EIGHT waves per CU, as two
// groups of four that alternate over the workgroup's 128 x 64 dW_0 tiles - one group runs a tile's K loop on the
// LDS-DMA ring (tile128_dma.h, NJ = 1) while the other applies RMSprop + EMA to the tile it finished before, from its
// own accumulator registers, and stores it. Included by pmlp_bwd.hip (inside its anonymous namespace).
//
// Why: in pmlp_fused_wgrad_kernel all 256 dW_0 tiles reach their epilogue together - 30 us of K loops with the memory
// system idle, then 10 us of optimiser traffic with the matrix pipes idle (timing-only builds: 53.0 us as is, 47.8
// without the epilogue, 32.1 without epilogue and small workgroups). A wave cannot overlap its own stores with its
// own next K loop (stores retire through the same in-order vmcnt as the DMA ring's loads), and a workgroup keeps its
// slot until its last store has issued, which is what the saturated memory pipe holds up; so the storing waves and
// the next loop's waves have to be resident together: two groups of one workgroup.
//
// s_barrier is workgroup-wide on gfx950: both groups execute the SAME barrier sequence. A K loop of H = 2 nch half
// chunks has H barriers (tile128_dma.h); the other group spreads its 32 units of work (load the state of a 32 x 32
// block in a loads-only phase, then update + store two rows per unit) over those H intervals, H a multiple of 32; one
// more barrier hands the ring over. The dW_i quadrants, the last layer and db_0 stay what they are in the tile kernel:
// small workgroups co-resident with the dW_0 work - which is why this kernel must fit 128 VGPRs (512-thread blocks,
// four waves per SIMD), and why the quadrants use the single-register-set routine here (wgrad_tile_B: 126 VGPRs; the
// tile_nt.h one needs 174).
//
// STATE: opt-in (NSVD_WGRAD_PP=1), correct (dW_0 bit-identical to the tile kernel; dW_i rounds differently, another
// routine), NOT faster: 53.2 us against 53.0 at cfg2. Timeline (stamps, us): tile 0's loops 0.6 -> 15.8-17.7 (as
// projected); tile 1's loops -> 35.2-42.3, i.e. 19-24 us instead of 16.6 - the other group's optimiser traffic and its
// barrier arrivals cost the loop what the hidden epilogue saved; the last tile's step, alone, -> 38.8-49.4; the dW_i
// quadrants crawl beside two back-to-back loops and end at 47.8-50.8. Without the small workgroups (timing only)
// this kernel takes 50.0 us where the tile kernel takes 45.5: overlapping half of the optimiser traffic with a K loop
// slows that loop - its operand stream shares the fabric, and every barrier waits for the storing group - by about
// what the hidden half saved. The 32 us of the timing-only build without any epilogue is therefore not what an
// overlap can reach; kept as the measured counter-example, not as a candidate.
#pragma once

constexpr int PP_THREADS = 512;

__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }

// the optimiser step on the group's 128 x 64 tile (blocks (0, 0) and (1, 0) of 32 x 32 per wave), in 32 units; SLOTS:
// `m` workgroup barriers behind every unit
template <bool EMA, bool SLOTS>
__device__ __forceinline__ void pp_epilogue(const WgradArgs& a, const f32x16 (&acc)[2][1], unsigned b0, unsigned ld,
                                            int hi, int m) {
    const NsvdOptPtrs& o = a.oW[0];
    float p[16], s[16], e[16];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
        // units 0-3: load block 0; 4-11: its rows, two per unit; 12-15: load block 1; 20-27: its rows (the gap lets
        // block 0's stores drain before the wait for block 1's loads, which also waits for them)
        if (u < 4 || (u >= 12 && u < 16)) {
            const int t = u >= 12 ? 1 : 0, q = u & 3;
#pragma unroll
            for (int r = 4 * q; r < 4 * q + 4; ++r) {
                const unsigned off = 4u * (b0 + 32u * t * ld + (unsigned)acc_row(r, hi) * ld);
                p[r] = wg_ld(o.p, off);
                s[r] = wg_ld(o.sq, off);
                e[r] = EMA ? wg_ld(o.ema, off) : 0.f;
            }
        } else if ((u >= 4 && u < 12) || (u >= 20 && u < 28)) {
            const int t = u >= 20 ? 1 : 0, q = u - (t ? 20 : 4);
#pragma unroll
            for (int r = 2 * q; r < 2 * q + 2; ++r) {
                const unsigned off = 4u * (b0 + 32u * t * ld + (unsigned)acc_row(r, hi) * ld);
                nsvd_rmsprop_upd(p[r], acc[t][0][r], s[r], e[r], EMA, a.h);
                wg_st(o.p, off, p[r]);
                wg_st(o.sq, off, s[r]);
                if (EMA) wg_st(o.ema, off, e[r]);
            }
        }
        if (SLOTS)
            for (int j = 0; j < m; ++j) pp_barrier();
    }
}

template <bool EMA>
__global__ void __launch_bounds__(PP_THREADS, 2) pmlp_wgrad_pp_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float smem_pp[4 * HID * A_LD];  // 72 KB: the ring, or a small tile's buffers
    static_assert(4 * HID * A_LD >= T128D_LDS_FLOATS, "the DMA ring must fit");
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= a.npp) {
        // small workgroups (4 waves of work; the block's other four leave at once)
        if (tid >= 256) return;
        const int bid = blockIdx.x - a.npp;
        WG_STAMP(0, bid < a.nB ? 2ull : 3ull);
        WG_STAMP(1, wall_clock64());
        if (bid < a.nB) wgrad_tile_B(a, smem_pp, smem_pp + 2 * HID * A_LD, bid, 0);
        else wgrad_tile_C(a, smem_pp, bid - a.nB, 0);
        WG_STAMP(6, wall_clock64());
        return;
    }
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int t = tid & 255, lane = t & 63, w = t >> 6;
    const int li = lane & 31, hi = lane >> 5, wm = w >> 1, wn = w & 1;
    const int ipw = a.ipw;  // tiles of this workgroup: adjacent feature tiles of one head
    int l, tg;
    xcd_block_map(blockIdx.x, a.hx, a.L, a.F / (64 * ipw), l, tg);
    const int nch = a.Bs / BK, H = 2 * nch, m = H / 32;
    f32x16 acc[2][1];
    unsigned b_mine = 0;
#ifdef NSVD_WG_STAMPS
#define PP_ST(slot) if (t == 0) g_wg_stamps[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64()
    if (tid == 0) g_wg_stamps[(size_t)blockIdx.x * 8] = 4ull;
#else
#define PP_ST(slot)
#endif
    for (int k = 0; k <= ipw; ++k) {
        const bool run = k < ipw && (k & 1) == grp;
        const bool epi = k > 0 && ((k - 1) & 1) == grp;
        if (k == ipw) {  // the last tile's step: nothing left to run beside it
            if (epi) pp_epilogue<EMA, false>(a, acc, b_mine, (unsigned)a.F, hi, 0);
            if (epi) PP_ST(7);
            break;
        }
        if (run) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
            PP_ST(1 + 2 * k);
            const int kf0 = (tg * ipw + k) * 64;
            const float* a_base = a.dz[0] + (size_t)l * HID * a.B;
            const float* b_base = a.phiTc + (size_t)kf0 * a.B;
            Tile128NoHook none;
            nsvd_tile128_dma<Tile128NoHook, 1>(a_base, b_base, (unsigned)a.B, (unsigned)a.B, nch, smem_pp, acc, none);
            b_mine = (unsigned)(((size_t)l * HID + 64 * wm) * a.F + kf0 + 32 * wn + li);
            PP_ST(2 + 2 * k);
        } else if (epi) {
            pp_epilogue<EMA, true>(a, acc, b_mine, (unsigned)a.F, hi, m);
            PP_ST(5 + k);
        } else {
            for (int j = 0; j < H; ++j) pp_barrier();
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the ring changes hands
    }
}

#undef PP_ST

// shapes this kernel takes: the fused step without stored gradients, no batch slices, K loops of a multiple of 16
// chunks, and an even number of 64-wide tiles per CU
inline int pp_items_per_wg(const nsvd_model_desc& d, int B, int S, int n_cu) {
    const int F = 2 * d.m;
    if (S != 1 || B % BK != 0 || (B / BK) % 16 != 0 || F % 64 != 0) return 0;
    const int items = (F / 64) * d.L;
    if (items % n_cu != 0) return 0;
    const int ipw = items / n_cu;
    if (ipw < 2 || (ipw & 1) || (F / 64) % ipw != 0) return 0;
    return ipw;
}
