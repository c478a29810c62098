// BACKWARD as one STREAMING kernel (pmlp_stream_bwd_kernel) for the plain-model shapes of the kernel-operator row
// (configs[3]: F = 2 m = 128 features, two hidden layers of 128, 64 heads x 8192 rows): included by pmlp_bwd.hip.
//
// The two-launch form (chain kernel + weight-gradient kernel) moves every dz_i through HBM - written by the chain
// (0.54 GB at configs[3]), read back with the activations by the weight-gradient tiles (HBM-side 1.1 + 2.0 GB per step,
// profiles/r03y_pmc_traffic_cfg4.txt) - and at this shape that traffic, not the matrix pipe, is what both kernels wait
// for (chain 340 us for 17 GFLOP, weight gradients 370 us for 34 GFLOP). Here a workgroup owns one head and one slice
// of the batch and walks it in chunks of 32 samples with W_1, dW_0 and dW_1 RESIDENT IN REGISTERS (64 + 2 x 64 of the
// lane's 512):
//   per chunk:  dz_1 = W_last dbase sigmoid'(a_1)                       registers (lane = sample, register = row)
//               dz_0 = (W_1^T dz_1) sigmoid'(a_0)                        64 MFMAs per wave, dz_1 through LDS
//               dW_1 += dz_1 a_0^T,  dW_0 += dz_0 phi^T                  2 x 64 MFMAs per wave, operands from LDS tiles
//               db_1, db_0, dW_last, db_last, d scales                   per-lane sums, reduced once at the end
// so the activations are read ONCE (0.54 GB), no dz is ever written, and the per-slice partial gradients go through
// the split-K second pass (wgrad_reduce_kernel: slices added in order, optimiser applied) that the two-launch form uses.
// d loss / d f comes from the caller (df) or from the EVD moments / partial moments exactly as in the chain kernel.
// Reference: autograd's backward of examples/models/mlp.py:204-221 + methods/nestedlora.py:98-111 (as pmlp_bwd.hip).
#pragma once

struct StreamArgs {
    const float* df;    // (B, ldl) or null (EVD mode)
    const float* jac;   // (B, ldl)
    const float* dsc;   // (B, ldl) or null
    const float* W1;    // (L, 128, 128)
    const float* Wl;    // (L, 128): the 128 -> 1 layer
    const float* a0;    // (L, 128, B) saved activations of layer 0
    const float* a1;    // (L, 128, B) of layer 1
    const float* phiTc; // (128, B)
    int B, L, ldl, l0;
    int S, Bs;          // batch slices, rows per slice (a multiple of 32)
    NsvdEvdIn evd;
    float* part;        // (S, part_stride) partial gradients, PartLayout offsets below
    size_t part_stride;
    size_t poW[3], pob[3], poscales;
    int dbg;  // diagnostic (NSVD_STREAM_DBG): 2 = operands requested for the first chunk only, 4 = block 0 prints the cycles
              // its wave 0 spent in each region of the loop
};

constexpr int SB_TILE = HID * A_LD;                  // one [128][36] tile
constexpr int SB_DZ = BS * H_LD;                     // [32][132]
constexpr int SB_MISC = 256 + 256 + 16;              // col[2 Lg <= 256] | dfp[8][32] | red
constexpr int SB_LDS_FLOATS = 4 * SB_TILE + SB_DZ + SB_MISC;  // 93 KB
constexpr size_t SB_LDS_BYTES = (size_t)SB_LDS_FLOATS * sizeof(float);

__device__ unsigned long long g_sb_stamps[8];  // NSVD_STREAM_DBG = 4: cycles per chunk and region (block 0, wave 0)

// workgroup barrier that does NOT drain the global loads in flight (__syncthreads() waits for vmcnt(0): the next chunk's
// operands, requested a chunk ahead, would be waited for at the very next barrier)
#define SB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ void __launch_bounds__(256, 1) pmlp_stream_bwd_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sb_smem[];
    float* A0s = sb_smem;                 // [128 k][36]   a_0 tile, samples contiguous
    float* PHs = A0s + SB_TILE;           // [128 kf][36]  phi^T tile
    float* DT1 = PHs + SB_TILE;           // [128 n][36]   dz_1, samples contiguous (each wave: its own 32 rows)
    float* DT0 = DT1 + SB_TILE;           // [128 n][36]   dz_0
    float* DZ = DT0 + SB_TILE;            // [32 c][132]   dz_1, rows contiguous (the chain's B operand)
    float* col = DZ + SB_DZ;              // [2][Lg] masked moment columns of this head
    float* dfp = col + 256;               // [8][32]
    float* red = dfp + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    // (head, slice): the workgroups of an XCD (block id mod 8) share a slice - its phi^T tiles and f rows - where S divides 8
    int l, slice;
    {
        const int bid = blockIdx.x;
        if (a.S <= 8 && 8 % a.S == 0 && a.L % (8 / a.S) == 0) {
            const int x = bid & 7, hpg = 8 / a.S;
            slice = x % a.S;
            l = (bid >> 3) * hpg + x / a.S;
        } else {
            l = bid / a.S;
            slice = bid - l * a.S;
        }
    }
    const int nch = a.Bs / BS;
    const int bbase = slice * a.Bs;
    const int Lg = a.df ? 0 : a.evd.Lg, lg = a.df ? 0 : a.evd.l_off + l;
    const int B1 = (a.B + 1) / 2, B2 = a.B - B1;

    // ---- resident operands: W_1 fragments of the chain step (W_1[n = 8 q + 4 hi + j][k = 32 w + li] at 4 q + j) and
    // this wave's rows of the 128 -> 1 layer
    float WF[64], wlv[16];
    {
        const float* Wi = a.W1 + (size_t)l * HID * HID + 32 * w + li;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float* wp = Wi + (size_t)(8 * q + 4 * hi) * HID;
            WF[4 * q] = wp[0];
            WF[4 * q + 1] = wp[HID];
            WF[4 * q + 2] = wp[2 * HID];
            WF[4 * q + 3] = wp[3 * HID];
        }
        const float* wl = a.Wl + (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) wlv[r] = wl[acc_row(r, hi)];
    }
    if (!a.df) {  // the masked moment columns of this head (chain kernel: the same expressions)
        for (int t = tid; t < 2 * Lg; t += 256) {
            const int h = t / Lg, lp = t - h * Lg;
            col[t] = nsvd_evd_mask_M(a.evd, lp, lg, Lg) * nsvd_evd_lam(a.evd, h, lp * Lg + lg, a.B, Lg);
        }
        if (blockIdx.x == 0) nsvd_evd_finish(a.evd, a.B, Lg, red);
    }

    f32x16 aW1[4], aW0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) aW1[j][r] = aW0[j][r] = 0.f;
    float db1[16], db0[16], dwl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) db1[r] = db0[r] = dwl[r] = 0.f;
    float dbl = 0.f, dscl = 0.f;

    // ---- the next chunk's operands, requested one chunk ahead: tile slabs (thread = row s_row + 32 k, float4 s_c4),
    // this lane's activations of layer 1, its row values, the row segment of f for the moment term
    const int s_row = tid >> 3, s_c4 = tid & 7;
    const int srow = tid & 31, sg = tid >> 5;
    const int seg = ((Lg + 31) >> 5) << 2;  // floats of f per thread (8 threads per row), a multiple of 4, <= 16
    const float* a0p = a.a0 + ((size_t)l * HID + s_row) * a.B + bbase + 4 * s_c4;
    const float* php = a.phiTc + (size_t)s_row * a.B + bbase + 4 * s_c4;
    const float* a1p = a.a1 + ((size_t)l * HID + 32 * w) * a.B + bbase + li;
    const size_t slab = (size_t)32 * a.B;
    float4 pa[4], pp[4], pf[4];
    float zn[16], tfn = 0.f, jacn = 0.f, dscn = 0.f, dfn = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) pa[k] = pp[k] = pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto request = [&](int c) {
        const int o = c * BS;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pa[k] = *reinterpret_cast<const float4*>(a0p + k * slab + o);
            pp[k] = *reinterpret_cast<const float4*>(php + k * slab + o);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) zn[r] = a1p[(size_t)acc_row(r, hi) * a.B + o];
        const int b = bbase + o + li;
        jacn = a.jac[(size_t)b * a.ldl + a.l0 + l];
        if (a.dsc) dscn = a.dsc[(size_t)b * a.ldl + a.l0 + l];
        if (a.df) {
            dfn = a.df[(size_t)b * a.ldl + a.l0 + l];
        } else {
            tfn = a.evd.Tf[(size_t)b * Lg + lg];
            const float* fr = a.evd.f + (size_t)(bbase + o + srow) * Lg + sg * seg;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * k < seg && sg * seg + 4 * k < Lg) pf[k] = *reinterpret_cast<const float4*>(fr + 4 * k);
        }
    };
    request(0);

    // Three barriers per chunk: (1) the previous chunk's readers are done -> tiles(c) and the moment dot go to LDS;
    // (2) tiles and dfp visible -> dz_1, dW_1 (this wave's own rows of DT1: no barrier needed); (3) DZ complete -> chain
    // step, dz_0, dW_0. The barriers are raw (no vmcnt(0)): the next chunk's operands, requested after barrier (1), have
    // the whole chunk to land. (Measured and dropped, round 5: the loop software-pipelined by one chunk - the weight-
    // gradient MFMAs of chunk c - 1 in the same region as the element-wise work of chunk c, via sched_group_barrier or
    // hand-placed behind every group of four MFMAs: no faster, 575 -> 580 / 813 us with the spills of the second form -
    // vector-ALU work of the SAME wave does not hide under its MFMAs on this part, scripts/experiments/README.md.)
    unsigned long long tr1 = 0, tr2 = 0, tr3 = 0, tba = 0, tbb = 0;
#define SB_T() (a.dbg & 4 ? __builtin_readcyclecounter() : 0ull)
    for (int c = 0; c < nch; ++c) {
        const unsigned long long t0 = SB_T();
        SB_BARRIER();  // (1)
        {
            float* A_ = A0s + s_row * A_LD + 4 * s_c4;
            float* P_ = PHs + s_row * A_LD + 4 * s_c4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                *reinterpret_cast<float4*>(A_ + k * 32 * A_LD) = pa[k];
                *reinterpret_cast<float4*>(P_ + k * 32 * A_LD) = pp[k];
            }
        }
        float zl[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) zl[r] = zn[r];
        const float tfv = tfn, jacv = jacn, dscv = dscn;
        float dfv = dfn;
        const int bcur = bbase + c * BS + li;
        if (!a.df) {
            // sum_l' f[b][l'] (M lam_other)[l'][l]: this thread's eighth of row srow; the halves of the batch take the
            // OTHER half's moments (reference methods/nestedlora.py:108-110)
            const float* cps = col + ((bbase + c * BS + srow) < B1 ? Lg : 0) + sg * seg;
            float part = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * k < seg && sg * seg + 4 * k < Lg) {
                    part = fmaf(pf[k].x, cps[4 * k], part);
                    part = fmaf(pf[k].y, cps[4 * k + 1], part);
                    part = fmaf(pf[k].z, cps[4 * k + 2], part);
                    part = fmaf(pf[k].w, cps[4 * k + 3], part);
                }
            dfp[sg * 32 + srow] = part;
        }
        if (c + 1 < nch && !(a.dbg & 2)) request(c + 1);
        const unsigned long long t1 = SB_T();
        SB_BARRIER();  // (2)
        const unsigned long long t2 = SB_T();
        if (!a.df) {
            const float* dq = dfp + li;
            const float acc = ((dq[0] + dq[32]) + (dq[64] + dq[96])) + ((dq[128] + dq[160]) + (dq[192] + dq[224]));
            const bool first = bcur < B1;
            dfv = a.evd.grad_scale * ((-4.f / (float)a.B) * nsvd_evd_mask_v(a.evd, lg, Lg) * tfv +
                                      (2.f / (float)(first ? B1 : B2)) * acc);
        }
        const float dbase = dfv * jacv;
        if (w == 0 && hi == 0) {
            dbl += dbase;
            dscl = fmaf(dfv, dscv, dscl);
        }
        float dz[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dz[r] = wlv[r] * dbase * nsvd_sigmoid_from_softplus(zl[r]);
            dwl[r] = fmaf(dbase, zl[r], dwl[r]);
            db1[r] += dz[r];
        }
        // dz_1 in both orientations: [c][n] for the chain step (every wave reads all 128 rows), [n][c] for dW_1 (this
        // wave's own rows: LDS operations of one wave complete in order)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(&DZ[li * H_LD + 32 * w + 8 * g + 4 * hi]) =
                make_float4(dz[4 * g], dz[4 * g + 1], dz[4 * g + 2], dz[4 * g + 3]);
#pragma unroll
        for (int r = 0; r < 16; ++r) DT1[(32 * w + acc_row(r, hi)) * A_LD + li] = dz[r];
        {   // dW_1 += dz_1 a_0^T: before the barrier, so that the waves' skew sits under these 64 MFMAs
            const float* Ap = DT1 + (32 * w + li) * A_LD + 4 * hi;
            const float* Bp = A0s + li * A_LD + 4 * hi;
            Frag<4> f0, f1;
            load_frag<4>(f0, Ap, Bp, A_LD);
            load_frag<4>(f1, Ap + 8, Bp + 8, A_LD);
            mma_frag<4>(aW1, f0);
            load_frag<4>(f0, Ap + 16, Bp + 16, A_LD);
            mma_frag<4>(aW1, f1);
            load_frag<4>(f1, Ap + 24, Bp + 24, A_LD);
            mma_frag<4>(aW1, f0);
            mma_frag<4>(aW1, f1);
        }
        const unsigned long long t3 = SB_T();
        SB_BARRIER();  // (3)
        const unsigned long long t4 = SB_T();
        {
            float a0v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) a0v[r] = A0s[(32 * w + acc_row(r, hi)) * A_LD + li];
            // the chain step, its B fragments one q-group ahead
            f32x16 acc1[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
            const float* Bq = DZ + li * H_LD + 4 * hi;
            Frag<1> g0, g1;
            g0.b[0] = *reinterpret_cast<const float4*>(Bq);
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                g1.b[0] = *reinterpret_cast<const float4*>(Bq + 8 * (q + 1));
                g0.a = make_float4(WF[4 * q], WF[4 * q + 1], WF[4 * q + 2], WF[4 * q + 3]);
                mma_frag<1>(acc1, g0);
                if (q + 2 < 16) g0.b[0] = *reinterpret_cast<const float4*>(Bq + 8 * (q + 2));
                g1.a = make_float4(WF[4 * q + 4], WF[4 * q + 5], WF[4 * q + 6], WF[4 * q + 7]);
                mma_frag<1>(acc1, g1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dz[r] = acc1[0][r] * nsvd_sigmoid_from_softplus(a0v[r]);
                db0[r] += dz[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) DT0[(32 * w + acc_row(r, hi)) * A_LD + li] = dz[r];
        {   // dW_0 += dz_0 phi^T (DT0: this wave's own rows)
            const float* Ap = DT0 + (32 * w + li) * A_LD + 4 * hi;
            const float* Bp = PHs + li * A_LD + 4 * hi;
            Frag<4> f0, f1;
            load_frag<4>(f0, Ap, Bp, A_LD);
            load_frag<4>(f1, Ap + 8, Bp + 8, A_LD);
            mma_frag<4>(aW0, f0);
            load_frag<4>(f0, Ap + 16, Bp + 16, A_LD);
            mma_frag<4>(aW0, f1);
            load_frag<4>(f1, Ap + 24, Bp + 24, A_LD);
            mma_frag<4>(aW0, f0);
            mma_frag<4>(aW0, f1);
        }
        const unsigned long long t5 = SB_T();
        tr1 += t1 - t0; tba += t2 - t1; tr2 += t3 - t2; tbb += t4 - t3; tr3 += t5 - t4;
    }
    if ((a.dbg & 4) && blockIdx.x == 0 && tid == 0) {
        g_sb_stamps[0] = tr1 / nch; g_sb_stamps[1] = tba / nch; g_sb_stamps[2] = tr2 / nch; g_sb_stamps[3] = tbb / nch;
        g_sb_stamps[4] = tr3 / nch; g_sb_stamps[5] = (unsigned long long)nch;
    }
#undef SB_T
#undef SB_BARRIER
    // ---- this slice's partial gradients
    float* P = a.part + (size_t)slice * a.part_stride;
    {
        float* g1 = P + a.poW[1] + ((size_t)l * HID + 32 * w) * HID + li;
        float* g0 = P + a.poW[0] + ((size_t)l * HID + 32 * w) * HID + li;  // (F = 128)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g1[(size_t)acc_row(r, hi) * HID + 32 * j] = aW1[j][r];
                g0[(size_t)acc_row(r, hi) * HID + 32 * j] = aW0[j][r];
            }
    }
    // row sums over the samples (lanes li of each half), in a fixed order
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            db1[r] += __shfl_xor(db1[r], off, 64);
            db0[r] += __shfl_xor(db0[r], off, 64);
            dwl[r] += __shfl_xor(dwl[r], off, 64);
        }
    }
    if (li == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t n = (size_t)l * HID + 32 * w + acc_row(r, hi);
            P[a.pob[1] + n] = db1[r];
            P[a.pob[0] + n] = db0[r];
            P[a.poW[2] + n] = dwl[r];
        }
    }
    if (w == 0) {
        dbl = nsvd_wave_sum(dbl);
        dscl = nsvd_wave_sum(dscl);
        if (lane == 0) {
            P[a.pob[2] + l] = dbl;
            if (a.dsc) P[a.poscales + l] = dscl;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same backward with the work of a chunk split over TWO wave groups per workgroup (8 waves, two per SIMD):
//   group A (waves 0-3)  the element-wise chain of chunk c: d loss / d f, dz_1, the chain step's 64 MFMAs, dz_0, the
//                        per-lane sums; W_1 in registers. Leaves dz_1 / dz_0 of its 32 rows in LDS (double buffered).
//   group B (waves 4-7)  the weight-gradient products of chunk c - 1: 128 MFMAs per wave on dW_1 / dW_0 (128 accumulator
//                        registers), and the staging of the a_0 / phi^T tiles they contract with.
// A single wave's vector-ALU work does not hide under its own MFMAs (the one-group kernel above: 20 K cycles per chunk
// for 12.3 K of MFMA issue); here B's MFMAs fill the matrix pipe while A's waves do theirs. s_barrier is workgroup-wide:
// both groups execute the same three barriers per iteration, and B's MFMAs are spread over the three segments.
constexpr int SB2_LDS_FLOATS = 6 * SB_TILE + SB_DZ + SB_MISC;  // a_0, phi^T tiles | DT1 x 2, DT0 x 2 | DZ | misc: 130 KB
constexpr size_t SB2_LDS_BYTES = (size_t)SB2_LDS_FLOATS * sizeof(float);

__global__ void __launch_bounds__(512, 1) pmlp_stream_bwd2_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sb_smem[];
    float* A0s = sb_smem;                 // [128 k][36]   a_0 tile of chunk c - 1 (group B)
    float* PHs = A0s + SB_TILE;           // [128 kf][36]  phi^T tile of chunk c - 1
    float* DT1 = PHs + SB_TILE;           // [2][128 n][36]  dz_1 (written by A for chunk c, read by B one iteration later)
    float* DT0 = DT1 + 2 * SB_TILE;       // [2][128 n][36]  dz_0
    float* DZ = DT0 + 2 * SB_TILE;        // [32 c][132]   dz_1, rows contiguous (A's chain step)
    float* col = DZ + SB_DZ;              // [2][Lg]
    float* dfp = col + 256;               // [8][32]
    const int grp = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    const int tid = threadIdx.x & 255;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    int l, slice;
    {
        const int bid = blockIdx.x;
        if (a.S <= 8 && 8 % a.S == 0 && a.L % (8 / a.S) == 0) {
            const int x = bid & 7, hpg = 8 / a.S;
            slice = x % a.S;
            l = (bid >> 3) * hpg + x / a.S;
        } else {
            l = bid / a.S;
            slice = bid - l * a.S;
        }
    }
    const int nch = a.Bs / BS;
    const int bbase = slice * a.Bs;
    float* P = a.part + (size_t)slice * a.part_stride;
#define SB_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    if (grp == 0) {
        // =============================================================================================== group A
        const int Lg = a.df ? 0 : a.evd.Lg, lg = a.df ? 0 : a.evd.l_off + l;
        const int B1 = (a.B + 1) / 2, B2 = a.B - B1;
        float WF[64], wlv[16];
        {
            const float* Wi = a.W1 + (size_t)l * HID * HID + 32 * w + li;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float* wp = Wi + (size_t)(8 * q + 4 * hi) * HID;
                WF[4 * q] = wp[0];
                WF[4 * q + 1] = wp[HID];
                WF[4 * q + 2] = wp[2 * HID];
                WF[4 * q + 3] = wp[3 * HID];
            }
            const float* wl = a.Wl + (size_t)l * HID + 32 * w;
#pragma unroll
            for (int r = 0; r < 16; ++r) wlv[r] = wl[acc_row(r, hi)];
        }
        if (!a.df) {
            for (int t = tid; t < 2 * Lg; t += 256) {
                const int h = t / Lg, lp = t - h * Lg;
                col[t] = nsvd_evd_mask_M(a.evd, lp, lg, Lg) * nsvd_evd_lam(a.evd, h, lp * Lg + lg, a.B, Lg);
            }
            // block 0: the loss scalars of the step (this group's 256 threads are the routine's; its one workgroup
            // barrier is matched by group B below)
            if (blockIdx.x == 0) nsvd_evd_finish(a.evd, a.B, Lg, dfp);
        }
        float dwl[16];  // (db_1 / db_0 are row sums of what group B reads anyway: taken there)
#pragma unroll
        for (int r = 0; r < 16; ++r) dwl[r] = 0.f;
        float dbl = 0.f, dscl = 0.f;
        const int srow = tid & 31, sg = tid >> 5;
        const int seg = ((Lg + 31) >> 5) << 2;
        const float* a1p = a.a1 + ((size_t)l * HID + 32 * w) * a.B + bbase + li;
        const float* a0p = a.a0 + ((size_t)l * HID + 32 * w) * a.B + bbase + li;
        float4 pf[4];
        float zn[16], a0n[16], tfn = 0.f, jacn = 0.f, dscn = 0.f, dfn = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) pf[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto request = [&](int c) {
            const int o = c * BS;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                zn[r] = a1p[(size_t)acc_row(r, hi) * a.B + o];
                a0n[r] = a0p[(size_t)acc_row(r, hi) * a.B + o];
            }
            const int b = bbase + o + li;
            jacn = a.jac[(size_t)b * a.ldl + a.l0 + l];
            if (a.dsc) dscn = a.dsc[(size_t)b * a.ldl + a.l0 + l];
            if (a.df) {
                dfn = a.df[(size_t)b * a.ldl + a.l0 + l];
            } else {
                tfn = a.evd.Tf[(size_t)b * Lg + lg];
                const float* fr = a.evd.f + (size_t)(bbase + o + srow) * Lg + sg * seg;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * k < seg && sg * seg + 4 * k < Lg) pf[k] = *reinterpret_cast<const float4*>(fr + 4 * k);
            }
        };
        request(0);
        SB_BARRIER();  // X0: the moment columns are in place (group B executes the same barrier)
        for (int c = 0; c <= nch; ++c) {
            const bool act = c < nch;
            // ---- segment 1: the moment dot of this chunk's rows
            if (act && !a.df) {
                const float* cps = col + ((bbase + c * BS + srow) < B1 ? Lg : 0) + sg * seg;
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * k < seg && sg * seg + 4 * k < Lg) {
                        part = fmaf(pf[k].x, cps[4 * k], part);
                        part = fmaf(pf[k].y, cps[4 * k + 1], part);
                        part = fmaf(pf[k].z, cps[4 * k + 2], part);
                        part = fmaf(pf[k].w, cps[4 * k + 3], part);
                    }
                dfp[sg * 32 + srow] = part;
            }
            SB_BARRIER();  // X1
            // ---- segment 2: dz_1
            float dz[16], s0[16];  // dz_1; sigmoid'(a_0) for the step after the chain (a_0's registers are then free)
            if (act) {
                float dfv = dfn;
                if (!a.df) {
                    const float* dq = dfp + li;
                    const float acc = ((dq[0] + dq[32]) + (dq[64] + dq[96])) + ((dq[128] + dq[160]) + (dq[192] + dq[224]));
                    const bool first = (bbase + c * BS + li) < B1;
                    dfv = a.evd.grad_scale * ((-4.f / (float)a.B) * nsvd_evd_mask_v(a.evd, lg, Lg) * tfn +
                                              (2.f / (float)(first ? B1 : B2)) * acc);
                }
                const float dbase = dfv * jacn;
                if (w == 0 && hi == 0) {
                    dbl += dbase;
                    dscl = fmaf(dfv, dscn, dscl);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s0[r] = nsvd_sigmoid_from_softplus(a0n[r]);
                    dz[r] = wlv[r] * dbase * nsvd_sigmoid_from_softplus(zn[r]);
                    dwl[r] = fmaf(dbase, zn[r], dwl[r]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(&DZ[li * H_LD + 32 * w + 8 * g + 4 * hi]) =
                        make_float4(dz[4 * g], dz[4 * g + 1], dz[4 * g + 2], dz[4 * g + 3]);
                float* D1 = DT1 + (c & 1) * SB_TILE;
#pragma unroll
                for (int r = 0; r < 16; ++r) D1[(32 * w + acc_row(r, hi)) * A_LD + li] = dz[r];
            }
            SB_BARRIER();  // X2
            // ---- segment 3: the next chunk's operands requested; chain step; dz_0
            if (act) {
                if (c + 1 < nch) request(c + 1);
                f32x16 acc1[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
                const float* Bq = DZ + li * H_LD + 4 * hi;
                Frag<1> g0, g1;
                g0.b[0] = *reinterpret_cast<const float4*>(Bq);
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    g1.b[0] = *reinterpret_cast<const float4*>(Bq + 8 * (q + 1));
                    g0.a = make_float4(WF[4 * q], WF[4 * q + 1], WF[4 * q + 2], WF[4 * q + 3]);
                    mma_frag<1>(acc1, g0);
                    if (q + 2 < 16) g0.b[0] = *reinterpret_cast<const float4*>(Bq + 8 * (q + 2));
                    g1.a = make_float4(WF[4 * q + 4], WF[4 * q + 5], WF[4 * q + 6], WF[4 * q + 7]);
                    mma_frag<1>(acc1, g1);
                }
                float* D0 = DT0 + (c & 1) * SB_TILE;
#pragma unroll
                for (int r = 0; r < 16; ++r) D0[(32 * w + acc_row(r, hi)) * A_LD + li] = acc1[0][r] * s0[r];
            }
            SB_BARRIER();  // X3
        }
        // ---- row sums over the samples, in a fixed order
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int off = 1; off < 32; off <<= 1) dwl[r] += __shfl_xor(dwl[r], off, 64);
        }
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) P[a.poW[2] + (size_t)l * HID + 32 * w + acc_row(r, hi)] = dwl[r];
        }
        if (w == 0) {
            dbl = nsvd_wave_sum(dbl);
            dscl = nsvd_wave_sum(dscl);
            if (lane == 0) {
                P[a.pob[2] + l] = dbl;
                if (a.dsc) P[a.poscales + l] = dscl;
            }
        }
    } else {
        // =============================================================================================== group B
        f32x16 aW1[4], aW0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) aW1[j][r] = aW0[j][r] = 0.f;
        const int s_row = tid >> 3, s_c4 = tid & 7;
        const float* a0p = a.a0 + ((size_t)l * HID + s_row) * a.B + bbase + 4 * s_c4;
        const float* php = a.phiTc + (size_t)s_row * a.B + bbase + 4 * s_c4;
        const size_t slab = (size_t)32 * a.B;
        float4 pa[4], pp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pa[k] = pp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        // db_1 / db_0: row sums of dz_1 / dz_0 over the samples - this lane's A fragments are row 32 w + li, four samples each
        float rs1 = 0.f, rs0 = 0.f;
#define SB2_RS(acc_, f_) acc_ += ((f_).a.x + (f_).a.y) + ((f_).a.z + (f_).a.w)
        if (blockIdx.x == 0 && !a.df) SB_BARRIER();  // (the barrier inside group A's nsvd_evd_finish)
        SB_BARRIER();  // X0
        for (int c = 0; c <= nch; ++c) {
            const bool act = c >= 1;  // this iteration multiplies chunk c - 1
            // ---- segment 1: the tiles of chunk c - 1 (requested an iteration ago) go to LDS - every wave of the group
            // has finished reading the previous ones (barrier X3)
            if (act) {
                float* A_ = A0s + s_row * A_LD + 4 * s_c4;
                float* P_ = PHs + s_row * A_LD + 4 * s_c4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    *reinterpret_cast<float4*>(A_ + k * 32 * A_LD) = pa[k];
                    *reinterpret_cast<float4*>(P_ + k * 32 * A_LD) = pp[k];
                }
            }
            SB_BARRIER();  // X1
            const float* D1 = DT1 + ((c + 1) & 1) * SB_TILE + (32 * w + li) * A_LD + 4 * hi;
            const float* D0 = DT0 + ((c + 1) & 1) * SB_TILE + (32 * w + li) * A_LD + 4 * hi;
            const float* Ba = A0s + li * A_LD + 4 * hi;
            const float* Bp = PHs + li * A_LD + 4 * hi;
            Frag<4> f0, f1;
            // ---- segment 2: the first half of dW_1
            if (act) {
                load_frag<4>(f0, D1, Ba, A_LD);
                load_frag<4>(f1, D1 + 8, Ba + 8, A_LD);
                mma_frag<4>(aW1, f0);
                SB2_RS(rs1, f0);
                mma_frag<4>(aW1, f1);
                SB2_RS(rs1, f1);
            }
            SB_BARRIER();  // X2
            // ---- segment 3: the tiles of chunk c requested; the rest of dW_1, dW_0
            if (c < nch) {
                const int o = c * BS;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pa[k] = *reinterpret_cast<const float4*>(a0p + k * slab + o);
                    pp[k] = *reinterpret_cast<const float4*>(php + k * slab + o);
                }
            }
            if (act) {
                load_frag<4>(f0, D1 + 16, Ba + 16, A_LD);
                load_frag<4>(f1, D1 + 24, Ba + 24, A_LD);
                mma_frag<4>(aW1, f0);
                SB2_RS(rs1, f0);
                load_frag<4>(f0, D0, Bp, A_LD);
                mma_frag<4>(aW1, f1);
                SB2_RS(rs1, f1);
                load_frag<4>(f1, D0 + 8, Bp + 8, A_LD);
                mma_frag<4>(aW0, f0);
                SB2_RS(rs0, f0);
                load_frag<4>(f0, D0 + 16, Bp + 16, A_LD);
                mma_frag<4>(aW0, f1);
                SB2_RS(rs0, f1);
                load_frag<4>(f1, D0 + 24, Bp + 24, A_LD);
                mma_frag<4>(aW0, f0);
                SB2_RS(rs0, f0);
                mma_frag<4>(aW0, f1);
                SB2_RS(rs0, f1);
            }
            SB_BARRIER();  // X3
        }
#undef SB2_RS
        float* g1 = P + a.poW[1] + ((size_t)l * HID + 32 * w) * HID + li;
        float* g0 = P + a.poW[0] + ((size_t)l * HID + 32 * w) * HID + li;  // (F = 128)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g1[(size_t)acc_row(r, hi) * HID + 32 * j] = aW1[j][r];
                g0[(size_t)acc_row(r, hi) * HID + 32 * j] = aW0[j][r];
            }
        rs1 += __shfl_xor(rs1, 32, 64);  // the two lane halves hold the two halves of every q-group's samples
        rs0 += __shfl_xor(rs0, 32, 64);
        if (hi == 0) {
            P[a.pob[1] + (size_t)l * HID + 32 * w + li] = rs1;
            P[a.pob[0] + (size_t)l * HID + 32 * w + li] = rs0;
        }
    }
#undef SB_BARRIER
}
