// Fused MFMA path, backward half: the data-gradient chain kernel, the weight-gradient kernel with the optimiser in
// its epilogue, and the split-K reduction (forward half and the overall design: pmlp_fwd.hip).
#include <stdlib.h>
#include "pmlp_common.h"
#include "evd_math.h"
#include "opt_math.h"
#include "tile_nt.h"
#include "tile128_dma.h"
#include "fourier_body.h"

using namespace nsvd_pmlp;

namespace {


// ================================================================================================
// BACKWARD, part 1 (pmlp_fused_bwd_chain_kernel): one workgroup = head l x 32 centre samples.
// Data gradients only, walking from the output back to layer 0 with the forward's transposed tiles
// (rows = hidden units, columns = samples on the lanes; wave w owns rows 32w..32w+31):
//   dz_{nh-1} = W_last * dbase * sigmoid(z_{nh-1}),   dbase = df * d f / d base
//   dz_{i-1}[k][c] = (sum_n W_i[n][k] dz_i[n][c]) * sigmoid(z_{i-1}[k][c])     64 MFMAs/wave, W_i from L2
// Every dz_i (L, 128, B) goes to the workspace; all weight/bias gradients are reductions over the
// batch and are done by pmlp_fused_wgrad_kernel without atomics.
// This is autograd's backward of reference mlp.py:204-221 for the centre evaluation only (the 2D
// shifted evaluations carry no gradient: nestedlora.py:108-111) and of pde/__init__.py:16.
struct ChainArgs {
    const float* df;             // (B, L) d loss / d f, or null: derive it from the moments below
    const float* jac;
    const float* dsc;            // null without the exponential mask
    const float* W[NSVD_MAX_LAYERS];
    const float* zsave[NSVD_MAX_LAYERS];
    float* dz[NSVD_MAX_LAYERS];  // (L, 128, B) per hidden layer
    float* dbase;                // (L, B)  df * d f / d base    (for the last-layer gradient)
    float* dfsc;                 // (L, B)  df * d f / d scales  (exponential mask only)
    int nlayers, B, L;           // L: heads of THIS launch (the pointers above start at its first head)
    int ldl, l0;                 // df, jac, dsc are (B, ldl) arrays of the whole model; this launch's heads start at l0
    // EVD-loss mode (df == null): NestedLoRALossFunctionEVD.backward evaluated per sample right here
    NsvdEvdIn evd;
    // guest workgroups (blockIdx >= chain_blocks): the NEXT batch's sampling + Fourier features into another workspace.
    // They depend on no weight and on nothing this launch computes; the chain blocks are latency-bound (one per CU at
    // cfg2, MFMA pipe 20 % busy), so the double-precision sincos work of the feature tiles rides in their shadow and
    // the stand-alone feature launch (5.7 us + a kernel boundary per step) disappears.
    int chain_blocks, feat_nx, feat_blocks;
    nsvd_feat::StencilArgs feat;
    // device-resident schedule (nsvd.h: nsvd_step_state), or null. One extra block (the last of the grid) derives the
    // step's learning rate / EMA decay from state->step for the weight-gradient kernel's optimiser epilogue, and the
    // guests' batch counter is feat.smp.offset + state->step (feat.smp.offset_add): nothing that changes from step to step is a launch
    // argument, so the captured launch replays along the schedule.
    nsvd_step_state* state;
    // direct mode only, or null: partial sums of the loss, added by the weight-gradient kernel in a fixed order:
    // [L][B / 32] operator term v_l sum_b f Tf over the workgroup's 32 rows, then [L] metric term
    // sum_l' M lam_f1 lam_f2 of column l (left by each head's first workgroup). The layout runs over ALL loss_L heads of
    // the model; this launch's heads start at l0 (head windows of one step: each window's chain fills its heads, the
    // LAST window's weight-gradient kernel adds them all)
    float* loss_part;
    int loss_L;
};

// PRE: the whole chain's weights and sigmoid inputs are fetched at kernel start (two or three hidden layers; ~290
// VGPRs, one workgroup per CU at a time). Pays only while the grid leaves CUs idle and the kernel is pure latency:
// measured steps/s with / without at cfg1 (64 workgroups) 4396 / 4260, cfg2 (256) 3778 / 3798, cfg3 per GPU (512)
// 3991 / 4184 - with every CU busy the 64 KB of W_i per workgroup are a throughput cost either way.
template <bool PRE>
__global__ void __launch_bounds__(256, 1) pmlp_fused_bwd_chain_kernel(ChainArgs a) {
    __shared__ __attribute__((aligned(16))) float DZ[BS * H_LD];  // [c][n]  n contiguous
    static_assert(BS * H_LD >= nsvd_feat::STAGE_FLOATS, "a feature tile's staging must fit the chain kernel's LDS");
    if ((int)blockIdx.x >= a.chain_blocks) {
        const int t = (int)blockIdx.x - a.chain_blocks;
        if (t >= a.feat_blocks) {  // the schedule block (only launched with a.state)
            if (threadIdx.x == 0) nsvd_step_state_derive(a.state);
            return;
        }
        const int by = t / a.feat_nx, bx = t - by * a.feat_nx;
        switch (a.feat.D) {  // (feat.smp.offset_add = &state->step: `step` is not written by this kernel)
            case 1: nsvd_feat::stencil_tile<1, 256>(a.feat, bx, by, DZ); break;
            case 2: nsvd_feat::stencil_tile<2, 256>(a.feat, bx, by, DZ); break;
            default: nsvd_feat::stencil_tile<3, 256>(a.feat, bx, by, DZ); break;
        }
        return;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int nsb = a.B / BS;
    const int l = blockIdx.x / nsb;
    const int b0 = (blockIdx.x - l * nsb) * BS;
    const int nh = a.nlayers - 1;
    const int b = b0 + li;
    const size_t row0 = ((size_t)l * HID + 32 * w) * a.B + b;

    // the last hidden layer's activations and the 128 -> 1 weights do not depend on d loss / d f: their loads fly
    // while the moments are taken
    float zl[16], wlv[16];
    {
        const float* zp = a.zsave[nh - 1] + row0;
        const float* wl = a.W[nh] + (size_t)l * HID + 32 * w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = acc_row(r, hi);
            zl[r] = zp[(size_t)n * a.B];
            wlv[r] = wl[n];
        }
    }
    // two or three hidden layers (every shipped configuration): the W_i fragments and sigmoid inputs of the whole
    // chain are fetched now as well - nothing below depends on them arriving, and a step of the chain is then 16 MFMAs
    // and one LDS exchange instead of a round trip to L2 per layer
    float wfa[64], wfb[64], zina[16], zinb[16];
    constexpr bool w1_ready = PRE;
    if (w1_ready) {
#define CHAIN_LOAD_TOP(i, WF, ZIN)                                                                      \
    {                                                                                                   \
        const float* zp = a.zsave[(i) - 1] + row0;                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) ZIN[r] = zp[(size_t)acc_row(r, hi) * a.B];       \
        const float* Wi = a.W[i] + (size_t)l * HID * HID + 32 * w + li;                                 \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                \
            const float* wp = Wi + (size_t)(8 * q + 4 * hi) * HID;                                      \
            WF[4 * q] = wp[0];                                                                          \
            WF[4 * q + 1] = wp[HID];                                                                    \
            WF[4 * q + 2] = wp[2 * HID];                                                                \
            WF[4 * q + 3] = wp[3 * HID];                                                                \
        }                                                                                               \
    }
        CHAIN_LOAD_TOP(nh - 1, wfa, zina)
        if (nh == 3) CHAIN_LOAD_TOP(1, wfb, zinb)
#undef CHAIN_LOAD_TOP
    }
    float dfv;
    if (a.df) {
        dfv = a.df[(size_t)b * a.ldl + a.l0 + l];
    } else {
        // d loss / d f[b][l] = gs * ( -(4/B) v_l Tf[b][l] + (2/B_half) sum_l' f[b][l'] M[l'][l] lam_other[l'][l] )
        // (reference methods/nestedlora.py:98-111 with f1, f2 = chunk(f, 2)); the moments are either the
        // reduced / all-reduced vector or this rank's per-chunk partial sums (reduced here, in a fixed order)
        float* col = DZ;  // [2][Lg] masked moment columns of (global) head lg (LDS scratch, free until the first exchange)
        float* red_mt = DZ + 256 + 1024;  // [4 waves] metric-term partials of the step's loss value (direct mode)
        const int Lg = a.evd.Lg, lg = a.evd.l_off + l;
        const int B1 = (a.B + 1) / 2, B2 = a.B - B1;
        if (a.evd.moments || a.evd.part) {
            for (int t = tid; t < 2 * Lg; t += 256) {
                const int h = t / Lg, lp = t - h * Lg;  // h = 0: lam_f1 column, 1: lam_f2 column
                col[t] = nsvd_evd_mask_M(a.evd, lp, lg, Lg) * nsvd_evd_lam(a.evd, h, lp * Lg + lg, a.B, Lg);
            }
            if (blockIdx.x == 0) nsvd_evd_finish(a.evd, a.B, Lg, DZ + 2 * Lg);
        } else {
            // direct mode: no moment kernel ran - the 2 Lg moments of THIS head's column straight from f.
            // Row-parallel: a thread takes whole rows of f (16 heads = one 64-byte read per row and half), multiplies
            // them by the row's f[.][lg] and keeps 2 x 16 partial moments; the 256 partial vectors are then summed by
            // a halving butterfly (32 shuffles instead of 32 x 6), across the waves through LDS, in a fixed order.
            // (One memory round trip and ~3 K cycles where 8 threads per moment walking the batch took ~8 K.)
            // The loss scalars are not produced.
            const int nchunk = (Lg + 15) >> 4;
            float* red = DZ + 256;  // [nchunk][4 waves][32]   (2 Lg <= 256 floats of col in front of it)
            const float* fh0 = a.evd.f;
            const float* fh1 = a.evd.f + (size_t)B1 * Lg;
            const bool vec = (Lg & 3) == 0;
            for (int c = 0; c < nchunk; ++c) {
                const int lp0 = 16 * c;
                float v[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) v[i] = 0.f;
                for (int r = tid; r < B1; r += 256) {
                    const bool has1 = r < B2;
                    const float* r0 = fh0 + (size_t)r * Lg;
                    const float* r1 = fh1 + (size_t)(has1 ? r : 0) * Lg;
                    float x0[16], x1[16];
                    const float g0 = r0[lg], g1 = has1 ? r1[lg] : 0.f;
                    if (vec && lp0 + 16 <= Lg) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 u0 = *reinterpret_cast<const float4*>(r0 + lp0 + 4 * q);
                            const float4 u1 = *reinterpret_cast<const float4*>(r1 + lp0 + 4 * q);
                            x0[4 * q] = u0.x; x0[4 * q + 1] = u0.y; x0[4 * q + 2] = u0.z; x0[4 * q + 3] = u0.w;
                            x1[4 * q] = u1.x; x1[4 * q + 1] = u1.y; x1[4 * q + 2] = u1.z; x1[4 * q + 3] = u1.w;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            x0[j] = lp0 + j < Lg ? r0[lp0 + j] : 0.f;
                            x1[j] = lp0 + j < Lg ? r1[lp0 + j] : 0.f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        v[j] = fmaf(x0[j], g0, v[j]);
                        v[16 + j] = fmaf(x1[j], g1, v[16 + j]);
                    }
                }
                // halving butterfly: after the step with partner lane ^ X a lane keeps half of its values, each now
                // the sum over both lanes; five steps leave ONE value per lane, the sum over its 32-lane half wave
#define NSVD_BFLY(N, X)                                                         \
    {                                                                           \
        const bool up = (lane & (X)) != 0;                                      \
        _Pragma("unroll") for (int k = 0; k < (N); ++k) {                       \
            const float send = up ? v[k] : v[k + (N)];                          \
            const float keep = up ? v[k + (N)] : v[k];                          \
            v[k] = keep + __shfl_xor(send, (X), 64);                            \
        }                                                                       \
    }
                NSVD_BFLY(16, 1) NSVD_BFLY(8, 2) NSVD_BFLY(4, 4) NSVD_BFLY(2, 8) NSVD_BFLY(1, 16)
#undef NSVD_BFLY
                const float tot = v[0] + __shfl_xor(v[0], 32, 64);
                // which of the 32 values this lane ended up with: bit 4 of the index <- lane bit 0, 3 <- 1, 2 <- 2, ...
                const int idx = ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
                if (lane < 32) red[(c * 4 + w) * 32 + idx] = tot;
            }
            __syncthreads();
            for (int t = tid; t < nchunk * 32; t += 256) {
                const int c = t >> 5, i = t & 31, h = i >> 4, lp = 16 * c + (i & 15);
                if (lp < Lg) {
                    const float* rp = red + c * 128 + i;
                    const float sum = (rp[0] + rp[32]) + (rp[64] + rp[96]);
                    col[h * Lg + lp] = nsvd_evd_mask_M(a.evd, lp, lg, Lg) * (sum / (float)(h ? B2 : B1));
                }
            }
            if (a.loss_part && b0 == 0) {
                // the metric term of this head's column, sum_l' (M lam_f1)[l'][l] lam_f2[l'][l] (reference:
                // methods/nestedlora.py:57-64): the thread of (h = 0, l') holds lam_f1, lane ^ 16 the lam_f2 of the same l'
                float mt = 0.f;
                for (int t = tid; t < ((nchunk * 32 + 63) & ~63); t += 256) {
                    const int c = t >> 5, i = t & 31, h = i >> 4, lp = 16 * c + (i & 15);
                    float lam = 0.f;
                    if (t < nchunk * 32 && lp < Lg) {
                        const float* rp = red + c * 128 + i;
                        lam = ((rp[0] + rp[32]) + (rp[64] + rp[96])) / (float)(h ? B2 : B1);
                    }
                    const float other = __shfl_xor(lam, 16, 64);
                    if (h == 0 && lp < Lg) mt = fmaf(nsvd_evd_mask_M(a.evd, lp, lg, Lg) * lam, other, mt);
                }
                mt = nsvd_wave_sum(mt);
                if (lane == 0) red_mt[w] = mt;
            }
        }
        __syncthreads();
        if (a.loss_part && b0 == 0 && tid == 0 && !(a.evd.moments || a.evd.part))
            a.loss_part[(size_t)a.loss_L * nsb + a.l0 + l] = (red_mt[0] + red_mt[1]) + (red_mt[2] + red_mt[3]);
        const bool first = b < B1;
        const float* fr = a.evd.f + (size_t)b * Lg;
        // sum_l' f[b][l'] (M lam_other)[l'][l] for the workgroup's 32 rows, cooperatively: thread t takes row t & 31 and
        // the t >> 5-th eighth of it (16-byte loads), the eight partial sums of a row meet in LDS and are added in a
        // fixed order. (Every thread walking its own whole row - 8 x redundant, Lg 4-byte loads each from 64 different
        // lines per instruction - cost the address unit more than the chain's MFMAs at L = 64.)
        float acc;
        {
            float* dfp = DZ + 1536;  // [8][32]  (col: DZ[0, 256), red: [256, 1280), red_mt: [1280, 1284))
            const int srow = tid & 31, sg = tid >> 5;
            const int bb = b0 + srow;
            const float* cps = col + (bb < B1 ? Lg : 0);  // the OTHER half's moments
            const float* frs = a.evd.f + (size_t)bb * Lg;
            const int seg = ((Lg + 31) >> 5) << 2;  // floats per eighth, a multiple of 4
            const int lp0 = sg * seg, lp1 = min(Lg, lp0 + seg);
            float part = 0.f;
            if ((Lg & 3) == 0) {
                for (int lp = lp0; lp < lp1; lp += 4) {
                    const float4 fv = *reinterpret_cast<const float4*>(frs + lp);
                    part = fmaf(fv.x, cps[lp], part);
                    part = fmaf(fv.y, cps[lp + 1], part);
                    part = fmaf(fv.z, cps[lp + 2], part);
                    part = fmaf(fv.w, cps[lp + 3], part);
                }
            } else {
                for (int lp = lp0; lp < lp1; ++lp) part = fmaf(frs[lp], cps[lp], part);
            }
            dfp[sg * 32 + srow] = part;
            __syncthreads();
            const float* dq = dfp + li;
            acc = ((dq[0] + dq[32]) + (dq[64] + dq[96])) + ((dq[128] + dq[160]) + (dq[192] + dq[224]));
        }
        const float tfv = a.evd.Tf[(size_t)b * Lg + lg];
        dfv = a.evd.grad_scale * ((-4.f / (float)a.B) * nsvd_evd_mask_v(a.evd, lg, Lg) * tfv +
                                  (2.f / (float)(first ? B1 : B2)) * acc);
        if (a.loss_part && w == 0) {
            // the operator term of the loss over this workgroup's 32 rows: v_l sum_b f[b][l] Tf[b][l]
            const float part = nsvd_wave_sum(hi == 0 ? fr[lg] * tfv : 0.f);
            if (lane == 0) a.loss_part[(size_t)(a.l0 + l) * nsb + (b0 / BS)] = nsvd_evd_mask_v(a.evd, lg, Lg) * part;
        }
        __syncthreads();  // col[] is dead before DZ is reused
    }
    const float dbase = dfv * a.jac[(size_t)b * a.ldl + a.l0 + l];
    if (w == 0 && hi == 0) {
        a.dbase[(size_t)l * a.B + b] = dbase;
        if (a.dfsc) a.dfsc[(size_t)l * a.B + b] = dfv * a.dsc[(size_t)b * a.ldl + a.l0 + l];
    }
    float dz[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dz[r] = wlv[r] * dbase * nsvd_sigmoid_from_softplus(zl[r]);
    // one step down the chain: store dz_i, exchange it through LDS, dz_{i-1} = (W_i^T dz_i) * sigmoid(z_{i-1}); the
    // W_i fragments (WF: 64 floats, W_i[n = 8 q + 4 hi + j][k = 32 w + li] at 4 q + j) and the sigmoid inputs (ZIN)
    // are in registers already
#define CHAIN_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define CHAIN_STEP(i, WF, ZIN)                                                                          \
    {                                                                                                   \
        float* o = a.dz[i] + row0;                                                                      \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) o[(size_t)acc_row(r, hi) * a.B] = dz[r];         \
        CHAIN_BARRIER(); /* previous round's LDS reads are done (raw: the dz stores above stay in flight) */ \
        _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                   \
            *reinterpret_cast<float4*>(&DZ[li * H_LD + 32 * w + 8 * g + 4 * hi]) =                      \
                make_float4(dz[4 * g], dz[4 * g + 1], dz[4 * g + 2], dz[4 * g + 3]);                    \
        CHAIN_BARRIER();                                                                                \
        f32x16 acc1[1];                                                                                 \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;                                \
        const float* Bp = DZ + li * H_LD + 4 * hi;                                                      \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                \
            Frag<1> f;                                                                                  \
            f.a = make_float4(WF[4 * q], WF[4 * q + 1], WF[4 * q + 2], WF[4 * q + 3]);                  \
            f.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * q);                                      \
            mma_frag<1>(acc1, f);                                                                       \
        }                                                                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                  \
            dz[r] = acc1[0][r] * nsvd_sigmoid_from_softplus(ZIN[r]);                                    \
    }
    if (PRE) {
        if (nh == 3) {
            CHAIN_STEP(2, wfa, zina)
            CHAIN_STEP(1, wfb, zinb)
        } else {
            CHAIN_STEP(1, wfa, zina)
        }
    } else {
        for (int i = nh - 1; i >= 1; --i) {
            float* o = a.dz[i] + row0;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)acc_row(r, hi) * a.B] = dz[r];
            // issue the loads the next tile needs before the LDS exchange: sigmoid inputs and W_i columns
            float zin[16];
            {
                const float* zp = a.zsave[i - 1] + row0;
#pragma unroll
                for (int r = 0; r < 16; ++r) zin[r] = zp[(size_t)acc_row(r, hi) * a.B];
            }
            CHAIN_BARRIER();  // previous round's LDS reads are done
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(&DZ[li * H_LD + 32 * w + 8 * g + 4 * hi]) =
                    make_float4(dz[4 * g], dz[4 * g + 1], dz[4 * g + 2], dz[4 * g + 3]);
            CHAIN_BARRIER();
            f32x16 acc1[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
            const float* Wi = a.W[i] + (size_t)l * HID * HID + 32 * w + li;  // W_i[n][k = 32w + li]
            const float* Bp = DZ + li * H_LD + 4 * hi;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                Frag<1> f;
                const float* wp = Wi + (size_t)(8 * q + 4 * hi) * HID;
                f.a = make_float4(wp[0], wp[HID], wp[2 * HID], wp[3 * HID]);
                f.b[0] = *reinterpret_cast<const float4*>(Bp + 8 * q);
                mma_frag<1>(acc1, f);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] = acc1[0][r] * nsvd_sigmoid_from_softplus(zin[r]);
        }
    }
#undef CHAIN_STEP
#undef CHAIN_BARRIER
    {
        float* o = a.dz[0] + row0;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[(size_t)acc_row(r, hi) * a.B] = dz[r];
    }
}

#include "pmlp_stream_bwd.h"

// ================================================================================================
// BACKWARD, part 2 (pmlp_fused_wgrad_kernel): every parameter gradient, one launch, no atomics.
// Three kinds of workgroup, told apart by blockIdx:
//   A  (F/128 * L):      dW_0[l][n][k] = sum_b dz_0[l][n][b] phi^T[k][b]: 128 x 128 tile, K = B, both
//                        operands b-contiguous, 4 waves as 2 x 2 of 64 x 64, operands by LDS-DMA through a ring
//                        of four half-chunk stages (tile128_dma.h).
//   B  (4 (nh-1) L):     dW_i[l][n][k] = sum_b dz_i[l][n][b] softplus(z_{i-1}[l][k][b]), i >= 1: one
//                        64 x 64 quadrant of the 128 x 128 result, 4 waves of one 32 x 32 tile; quadrants
//                        in column 0 also write db_i. (128 x 64 half tiles were measured slower: fewer,
//                        longer workgroups next to the dW_0 tiles.)
//   C  (4 L):            dW_last[l][n] = sum_b dbase[b] softplus(z_{nh-1}[l][n][b]), db_last, d scales, and
//                        db_0[l][n] = sum_b dz_0[l][n][b] for the same 32 rows.
// Timeline at cfg2 (NSVD_WG_STAMPS build, scripts/dev/wgrad_stamps.py), fused optimiser step: the A tiles run
// their K loop in 70-79 K cycles (65.5 K of MFMA issue; 69.7 K with nothing else on the chip) and their
// epilogue - RMSprop + EMA on the tile, 112 MB through HBM for the whole kernel, the state of half of each tile
// already in registers (fetched under the last two chunks of the loop) - in 22-26 K, ending at 38-48 us; the B
// tiles, co-resident with them from t = 0, end at 46-49 us, the C tiles at 42-45 us; kernel 52 us in rocprof
// (round 1: 60 us; separate backward and optimiser launches 50 + 20.5 us). Raising the B / C wave priority
// (s_setprio 3) shortens them but stretches the A loops by the same amount, raising the A priority changes
// nothing: no gain either way. Nor does cutting the B quadrants into two batch halves (twice the workgroups, one per
// CU, the half arriving second adds its partner's partial): they are not shorter - next to a dW_0 tile a small
// workgroup advances at the pace the tile leaves it, whatever its own work - and every dW_0 loop is slowed instead of
// half of them: 54.6 us against 52.9. Holding the small workgroups back (a counter of dW_0 tiles past their K loop,
// bounded wait) until 50 / 90 / 100 % of the loops are done, with whole or halved quadrants: 272-275 us per step
// against 263 - the dW_0 loops do get the CU to themselves, but the small workgroups' work then lands on the
// memory-bound epilogues instead of under the loops.
struct WgradArgs {
    const float* dz[NSVD_MAX_LAYERS];     // (L, 128, B)
    const float* zsave[NSVD_MAX_LAYERS];  // (L, 128, B)
    const float* phiTc;                   // (F, B)
    const float* dbase;                   // (L, B) from the chain kernel
    const float* dfsc;                    // (L, B), null without the exponential mask
    float* gW[NSVD_MAX_LAYERS];           // gradients; may be null when the optimiser step is fused (opt != 0)
    float* gb[NSVD_MAX_LAYERS];
    float* gscales;
    int nlayers, B, L, F;
    int nA, nB;
    int hx;  // XCD map of the dW_0 tiles (pick_xcd_remap_wgrad; 0: none)
    int tw;  // features per dW_0 tile: 128, or 64 when that many tiles would leave half the CUs without one (S == 1)
    int bid0;  // first logical block of this launch (the A tiles and the B/C tiles are launched separately)
    // fused RMSprop + EMA epilogue (opt != 0): every gradient element is applied to its parameter in place
    // the moment it leaves the accumulator, so it never makes the HBM round trip (-8 B/parameter, -1 launch).
    // Safe in place: this kernel reads no parameter, the chain kernel that does has already run.
    int opt;
    NsvdHyper h;
    NsvdOptPtrs oW[NSVD_MAX_LAYERS], ob[NSVD_MAX_LAYERS], oscales;
    // split-K over the batch (S > 1; head-parallel ranks own few heads on many rows, so the tile count alone
    // would not fill the chip): tile (unit, slice) contracts rows [slice * Bs, (slice + 1) * Bs) and stores a
    // partial gradient into slice `slice` of `part`; wgrad_reduce_kernel adds the slices in order and stores the
    // gradient / takes the optimiser step.
    int S, Bs;
    float* part;
    size_t part_stride;                                        // floats per slice
    size_t poW[NSVD_MAX_LAYERS], pob[NSVD_MAX_LAYERS], poscales;  // tensor offsets inside a slice
    // device-resident schedule (or null): `h` is read from state->cur (written by the chain kernel's schedule block)
    // and ONE thread of the last kernel of the step increments state->step (state; null in a window of a step that is
    // not the step's last: hstate is then where the optimiser scalars are read)
    nsvd_step_state* state;
    nsvd_step_state* hstate;
    // the step's loss from the chain kernel's per-head partials (direct mode), or null
    const float* loss_part;  // [loss_L][B / 32] operator-term partials, then [loss_L] metric-term partials
    float* loss;             // {loss, operator term, metric term}
    int loss_L;
    // EMIT instantiations (bf16x3 steps): the three bf16 planes of every UPDATED element of W_0 / W_1 .. in the
    // fragment-major layouts the bf16x3 forward reads (pmlp_layer0_bf3.h: w0_split_kernel's), so that the step needs no
    // split launch (9 us, 17 MB read + 26 MB written per step): same rounding, same bits
    unsigned short* w0p;
    unsigned short* whp;
};

// where a gradient element goes: the caller's gradient tensor (+ fused optimiser) or this slice's partial buffer
struct WgDst {
    float* g;
    int opt;
};
__device__ __forceinline__ WgDst wg_dst(const WgradArgs& a, float* g, size_t part_off, int slice) {
    if (a.S > 1) return WgDst{a.part + (size_t)slice * a.part_stride + part_off, 0};
    return WgDst{g, a.opt};
}

#ifdef NSVD_WG_STAMPS
// diagnostic build: per-block (kind, realtime start/end, cycles in prologue / loop / epilogue)
__device__ unsigned long long g_wg_stamps[1024 * 8];
#define WG_STAMP(slot, v) if (threadIdx.x == 0) g_wg_stamps[(size_t)blockIdx.x * 8 + (slot)] = (v)
#else
#define WG_STAMP(slot, v)
#endif

// one gradient element: store it and / or take the optimiser step on its parameter
__device__ __forceinline__ void wg_emit1(const NsvdHyper& h, const WgDst& d, const NsvdOptPtrs& o, size_t off,
                                         float val) {
    float* g = d.g;
    if (g) g[off] = val;
    if (d.opt) {
        float pv = o.p[off], sv = o.sq[off], ev = o.ema ? o.ema[off] : 0.f;
        nsvd_rmsprop_upd(pv, val, sv, ev, o.ema != nullptr, h);
        o.p[off] = pv;
        o.sq[off] = sv;
        if (o.ema) o.ema[off] = ev;
    }
}

// the 16 accumulator registers of one 32 x 32 MFMA tile: rows acc_row(r, hi) * ld, this lane's column at `base`.
// Element offsets are 32-bit (every tensor is far below 2^32 bytes: checked on the host) so that the three state
// arrays share one offset register per element and the loads take the scalar-base form; with 64-bit offsets the 48
// loads in flight spill, and every spill waits for its load.
__device__ __forceinline__ float wg_ld(const float* p, unsigned byte_off) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + byte_off);
}
__device__ __forceinline__ void wg_st(float* p, unsigned byte_off, float v) {
    *reinterpret_cast<float*>(reinterpret_cast<char*>(p) + byte_off) = v;
}

// bf16 planes of one updated 32 x 32 block (EMIT): this lane holds column k = li of the block for the 16 rows
// acc_row(r, hi); in both fragment-major layouts the 16-byte item of (row, 8 consecutive k) is 16 (row) bytes from the
// block's base, the k-octet (k / 8) selects k-step and lane half, the planes are PLANE bytes apart: two bytes per
// (row, plane) from this lane, eight lanes complete an item, 8 items (128 B) per instruction and wave half.
struct WgPlanes {
    unsigned short* base;  // null: no emission
    unsigned lane_off;     // bytes: this lane's k inside the block
    unsigned plane;        // bytes between planes
};
// W_0 block: head l, hidden rows 32 wr .. (wr = 0..3), features kabs0 .. kabs0 + 31 (a multiple of 32)
__device__ __forceinline__ WgPlanes wg_planes_w0(unsigned short* w0p, int l, int wr, int kabs0, int m, int li) {
    WgPlanes p;
    p.base = nullptr; p.lane_off = 0; p.plane = 2048;
    if (!w0p) return p;
    const int nch = 2 * m / 32;
    const int c = kabs0 < m ? 2 * (kabs0 / 32) : 2 * ((kabs0 - m) / 32) + 1;  // pair-of-chunks order of the K loop
    // P[((((l nch + c) 4 + wr) 3 + p) 2 + ks) 64 + 32 hi' + row] x 16 B, ks = k / 16, hi' = (k / 8) & 1
    p.base = w0p + ((size_t)(l * nch + c) * 4 + wr) * (6 * 64 * 8);
    p.lane_off = (unsigned)(((li >> 4) * 64 + ((li >> 3) & 1) * 32) * 16 + 2 * (li & 7));
    return p;
}
// W_j block (j >= 1; jh = j - 1): head l of L, rows 32 wr .., columns kabs0 .. + 31
__device__ __forceinline__ WgPlanes wg_planes_wh(unsigned short* whp, int jh, int L, int l, int wr, int kabs0, int li) {
    WgPlanes p;
    p.base = nullptr; p.lane_off = 0; p.plane = 1024;
    if (!whp) return p;
    // Ph[((((jh L + l) 4 + wr) 8 + ks) 3 + p) 64 + 32 hi' + row] x 16 B, ks = k / 16 of the 128 columns
    const int k = kabs0 + li;
    p.base = whp + ((size_t)(jh * L + l) * 4 + wr) * (8 * 3 * 64 * 8);
    p.lane_off = (unsigned)(((k >> 4) * 3 * 64 + ((k >> 3) & 1) * 32) * 16 + 2 * (k & 7));
    return p;
}
__device__ __forceinline__ void wg_planes_store(const WgPlanes& pl, int r, int hi, float v) {
    // three-way split by round-to-nearest residuals (nsvd_bf3_split's arithmetic, one value)
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const unsigned h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{v, 0.f}, b2_t)) & 0xffffu;
    const float r1 = v - __uint_as_float(h0 << 16);
    const unsigned h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{r1, 0.f}, b2_t)) & 0xffffu;
    const float r2 = r1 - __uint_as_float(h1 << 16);
    const unsigned h2 = __builtin_bit_cast(unsigned, __builtin_convertvector(f2_t{r2, 0.f}, b2_t)) & 0xffffu;
    char* q = reinterpret_cast<char*>(pl.base) + pl.lane_off + 16u * (unsigned)acc_row(r, hi);
    *reinterpret_cast<unsigned short*>(q) = (unsigned short)h0;
    *reinterpret_cast<unsigned short*>(q + pl.plane) = (unsigned short)h1;
    *reinterpret_cast<unsigned short*>(q + 2 * pl.plane) = (unsigned short)h2;
}

template <bool EMA>
__device__ __forceinline__ void wg_opt16(const NsvdHyper& h, const NsvdOptPtrs& o, unsigned base, unsigned ld, int hi,
                                         const f32x16& acc, const WgPlanes pl = WgPlanes{nullptr, 0, 0}) {
    float pv[16], sv[16], ev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // 48 independent loads in flight
        const unsigned off = 4u * (base + (unsigned)acc_row(r, hi) * ld);
        pv[r] = wg_ld(o.p, off);
        sv[r] = wg_ld(o.sq, off);
        ev[r] = EMA ? wg_ld(o.ema, off) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = 4u * (base + (unsigned)acc_row(r, hi) * ld);
        nsvd_rmsprop_upd(pv[r], acc[r], sv[r], ev[r], EMA, h);
        wg_st(o.p, off, pv[r]);
        wg_st(o.sq, off, sv[r]);
        if (EMA) wg_st(o.ema, off, ev[r]);
        if (pl.base) wg_planes_store(pl, r, hi, pv[r]);
    }
}

__device__ __forceinline__ void wg_emit16(const NsvdHyper& h, const WgDst& d, const NsvdOptPtrs& o, size_t base,
                                          size_t ld, int hi, const f32x16& acc, const WgPlanes pl = WgPlanes{nullptr, 0, 0}) {
    float* g = d.g;
    if (g) {
#pragma unroll
        for (int r = 0; r < 16; ++r) wg_st(g, 4u * ((unsigned)base + (unsigned)acc_row(r, hi) * (unsigned)ld), acc[r]);
    }
    if (!d.opt) return;
    if (o.ema) wg_opt16<true>(h, o, (unsigned)base, (unsigned)ld, hi, acc, pl);
    else wg_opt16<false>(h, o, (unsigned)base, (unsigned)ld, hi, acc, pl);
}

// stage one 32-row x 32-column (float4 per thread) slab global -> registers
#define WG_LD(dst, src) dst = *reinterpret_cast<const float4*>(src)
#define WG_ST(dst, v) *reinterpret_cast<float4*>(dst) = (v)

__device__ __forceinline__ float4 softplus4(float4 v) {
    return make_float4(nsvd_softplus(v.x), nsvd_softplus(v.y), nsvd_softplus(v.z), nsvd_softplus(v.w));
}

// Optimiser state (parameter, square average, EMA shadow) of the tile's blocks acc[0][0] and acc[0][1], fetched under
// the last four half chunks of the K loop (the hook of tile128_dma.h): the epilogue used to start with every wave of the
// chip asking for its state at once and four serial load -> update -> store round trips per wave.
template <bool EMA, int NJ>
struct OptPrefetch {
    static constexpr int LOADS = EMA ? 3 : 2;  // per issue<K, PART>: one row of the block
    const float *P, *S, *E;
    unsigned base, ld;  // element offset of block (0, 0) at this lane's column; row pitch
    int hi;
    float p[2][16], s[2][16], e[2][16];
    template <int K, int PART>
    __device__ __forceinline__ void issue() {  // K = 0..7: rows 4 (K % 4) .. + 3 of prefetched block K / 4; PART: which
        constexpr int t = K / 4, r = 4 * (K % 4) + PART;
        // the second prefetched block: (0, 1), or (1, 0) when the tile has no (0, 1)
        const unsigned off = 4u * (base + (NJ == 2 ? 32u * t : 32u * t * ld) + (unsigned)acc_row(r, hi) * ld);
        p[t][r] = wg_ld(P, off);
        s[t][r] = wg_ld(S, off);
        if (EMA) e[t][r] = wg_ld(E, off);
    }
};

template <bool EMA>
__device__ __forceinline__ void wg_opt16_load(const NsvdOptPtrs& o, unsigned base, unsigned ld, int hi,
                                              float (&pv)[16], float (&sv)[16], float (&ev)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = 4u * (base + (unsigned)acc_row(r, hi) * ld);
        pv[r] = wg_ld(o.p, off);
        sv[r] = wg_ld(o.sq, off);
        ev[r] = EMA ? wg_ld(o.ema, off) : 0.f;
    }
}

template <bool EMA>
__device__ __forceinline__ void wg_opt16_apply(const NsvdHyper& h, const NsvdOptPtrs& o, unsigned base, unsigned ld,
                                               int hi, const f32x16& acc, float (&pv)[16], float (&sv)[16],
                                               float (&ev)[16], const WgPlanes pl = WgPlanes{nullptr, 0, 0}) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned off = 4u * (base + (unsigned)acc_row(r, hi) * ld);
        if (!EMA) ev[r] = 0.f;
        nsvd_rmsprop_upd(pv[r], acc[r], sv[r], ev[r], EMA, h);
        wg_st(o.p, off, pv[r]);
        wg_st(o.sq, off, sv[r]);
        if (EMA) wg_st(o.ema, off, ev[r]);
        if (pl.base) wg_planes_store(pl, r, hi, pv[r]);
    }
}

// MODE 0: gradients stored (or split-K partials); 1 / 2: optimiser step in the epilogue without / with the EMA
// shadow, the state of two blocks prefetched under the K loop (needs the pipelined loop, i.e. >= 4 chunks, and no
// gradient output). NJ = 2: 128 x 128 tile (hidden units x features); NJ = 1: 128 x 64, chosen by the host when the
// 128-wide tiles would leave half the CUs without one. The bias gradient db_0 is taken by the C workgroups.
template <int MODE, int NJ, bool EMIT = false>
__device__ __forceinline__ void wgrad_tile_A(const WgradArgs& a, const NsvdHyper& h, float* lds, int unit, int slice) {
    constexpr int TW = 64 * NJ;  // features per tile
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nkt = a.F / TW;
    const int l = unit / nkt;
    const int kf0 = (unit - l * nkt) * TW;

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* a_base = a.dz[0] + (size_t)l * HID * a.B + (size_t)slice * a.Bs;
    const float* b_base = a.phiTc + (size_t)kf0 * a.B + (size_t)slice * a.Bs;
    const size_t o = ((size_t)l * HID + 64 * wm) * a.F + kf0 + 32 * NJ * wn + li;
    WG_STAMP(0, 1ull);
    WG_STAMP(1, wall_clock64());
    WG_STAMP(2, __builtin_readcyclecounter());
    if (MODE == 0) {
        nsvd_tile128_dma(a_base, b_base, (unsigned)a.B, (unsigned)a.B, a.Bs / BK, lds, acc);
        WG_STAMP(4, __builtin_readcyclecounter());
        const WgDst dW = wg_dst(a, a.gW[0], a.poW[0], slice);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                wg_emit16(h, dW, a.oW[0], o + (size_t)(32 * i) * a.F + 32 * j, a.F, hi, acc[i][j],
                          wg_planes_w0(EMIT ? a.w0p : nullptr, l, 2 * wm + i, kf0 + 32 * NJ * wn + 32 * j, a.F / 2, li));
    } else {
        constexpr bool EMA = MODE == 2;
        const NsvdOptPtrs& op = a.oW[0];
        const unsigned ld = (unsigned)a.F, b00 = (unsigned)o, b10 = b00 + 32u * ld;
        OptPrefetch<EMA, NJ> pf;
        pf.P = op.p; pf.S = op.sq; pf.E = op.ema;
        pf.base = b00; pf.ld = ld; pf.hi = hi;
        nsvd_tile128_dma(a_base, b_base, (unsigned)a.B, (unsigned)a.B, a.Bs / BK, lds, acc, pf);
        WG_STAMP(4, __builtin_readcyclecounter());
        if (NJ == 2) {
            // blocks (0,0), (0,1) have their state; the loads of (1,0), (1,1) go out behind the stores of the former
            float p2[16], s2[16], e2[16], p3[16], s3[16], e3[16];
#define WG_PL(i_, j_) wg_planes_w0(EMIT ? a.w0p : nullptr, l, 2 * wm + (i_), kf0 + 32 * NJ * wn + 32 * (j_), a.F / 2, li)
            wg_opt16_apply<EMA>(h, op, b00, ld, hi, acc[0][0], pf.p[0], pf.s[0], pf.e[0], WG_PL(0, 0));
            wg_opt16_load<EMA>(op, b10, ld, hi, p2, s2, e2);
            wg_opt16_apply<EMA>(h, op, b00 + 32u, ld, hi, acc[0][NJ - 1], pf.p[1], pf.s[1], pf.e[1], WG_PL(0, NJ - 1));
            wg_opt16_load<EMA>(op, b10 + 32u, ld, hi, p3, s3, e3);
            wg_opt16_apply<EMA>(h, op, b10, ld, hi, acc[1][0], p2, s2, e2, WG_PL(1, 0));
            wg_opt16_apply<EMA>(h, op, b10 + 32u, ld, hi, acc[1][NJ - 1], p3, s3, e3, WG_PL(1, NJ - 1));
        } else {
            // both blocks of the tile have their state
            wg_opt16_apply<EMA>(h, op, b00, ld, hi, acc[0][0], pf.p[0], pf.s[0], pf.e[0], WG_PL(0, 0));
            wg_opt16_apply<EMA>(h, op, b10, ld, hi, acc[1][0], pf.p[1], pf.s[1], pf.e[1], WG_PL(1, 0));
#undef WG_PL
        }
    }
    WG_STAMP(5, __builtin_readcyclecounter());
    WG_STAMP(6, wall_clock64());
}

// dW_i quadrant through the shared C = A B^T tile routine (tile_nt.h): both operands are plain (L, 128, B) rows now
// that the forward saves activations - no softplus while staging, loads two chunks ahead, four accumulator chains.
// Needs the slice length to be a multiple of 64 (the 32-chunk form below takes the rest).
template <bool EMIT = false>
__device__ __forceinline__ void wgrad_tile_B64(const WgradArgs& a, const NsvdHyper& h, float* lds, int unit, int slice) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int quad = unit & 3;
    const int rest = unit >> 2;
    const int l = rest % a.L;
    const int i = 1 + rest / a.L;
    const int n0 = (quad >> 1) * 64, k0 = (quad & 1) * 64;
    const float* A = a.dz[i] + ((size_t)l * HID + n0) * a.B;
    const float* Bm = a.zsave[i - 1] + ((size_t)l * HID + k0) * a.B;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
    const int b0 = slice * a.Bs;
    if (k0 == 0) nsvd_tile_nt<true>(A, a.B, Bm, a.B, b0, b0 + a.Bs, lds, acc, rs);
    else nsvd_tile_nt<false>(A, a.B, Bm, a.B, b0, b0 + a.Bs, lds, acc, rs);
    wg_emit16(h, wg_dst(a, a.gW[i], a.poW[i], slice), a.oW[i],
              ((size_t)l * HID + n0 + 32 * (wv & 1)) * HID + k0 + 32 * (wv >> 1) + li, HID, hi, acc,
              wg_planes_wh(EMIT ? a.whp : nullptr, i - 1, a.L, l, n0 / 32 + (wv & 1), k0 + 32 * (wv >> 1), li));
    if (k0 == 0) {
        // bias gradient: the 16 threads t & 15 of a staging row hold partial sums of rows (t >> 4) + 16 j
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) rs[j] += __shfl_xor(rs[j], off, 64);
        if ((tid & 15) == 0) {
            const WgDst db = wg_dst(a, a.gb[i], a.pob[i], slice);
            const size_t gb = (size_t)l * HID + n0 + (tid >> 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) wg_emit1(h, db, a.ob[i], gb + 16 * j, rs[j]);
        }
    }
}

template <bool EMIT = false>
__device__ __forceinline__ void wgrad_tile_B(const WgradArgs& a, const NsvdHyper& h, float* As, float* Bs, int unit, int slice) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int li = lane & 31, hi = lane >> 5;
    const int wm = w >> 1, wn = w & 1;
    const int nh = a.nlayers - 1;
    // unit -> (layer i in 1..nh-1, head l, quadrant)
    const int quad = unit & 3;
    const int rest = unit >> 2;
    const int l = rest % a.L;
    const int i = 1 + rest / a.L;
    (void)nh;
    const int n0 = (quad >> 1) * 64, k0 = (quad & 1) * 64;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int s_row = tid >> 3, s_c4 = tid & 7;  // 32 rows x 8 float4; two slabs per operand (64 rows)
    const float* a_src = a.dz[i] + ((size_t)l * HID + n0 + s_row) * a.B + (size_t)slice * a.Bs + 4 * s_c4;
    const float* b_src = a.zsave[i - 1] + ((size_t)l * HID + k0 + s_row) * a.B + (size_t)slice * a.Bs + 4 * s_c4;
    const size_t step = (size_t)32 * a.B;
    float4 pa0, pa1, pb0, pb1, qa0, qa1, qb0, qb1;
    float rs0 = 0.f, rs1 = 0.f;
    constexpr int TS = 64 * A_LD;  // one 64-row tile
#define WB_LOAD(S, c)                                 \
    {                                                 \
        WG_LD(S##a0, a_src + (c) * BK);               \
        WG_LD(S##a1, a_src + (c) * BK + step);        \
        WG_LD(S##b0, b_src + (c) * BK);               \
        WG_LD(S##b1, b_src + (c) * BK + step);        \
    }
#define WB_STORE(S, buf)                                                \
    {                                                                   \
        float* Ab_ = As + (buf) * TS + s_row * A_LD + 4 * s_c4;         \
        float* Bb_ = Bs + (buf) * TS + s_row * A_LD + 4 * s_c4;         \
        WG_ST(Ab_, S##a0);                                              \
        WG_ST(Ab_ + 32 * A_LD, S##a1);                                  \
        WG_ST(Bb_, S##b0);                                              \
        WG_ST(Bb_ + 32 * A_LD, S##b1);                                  \
        rs0 += (S##a0.x + S##a0.y) + (S##a0.z + S##a0.w);               \
        rs1 += (S##a1.x + S##a1.y) + (S##a1.z + S##a1.w);               \
    }
#define WB_COMPUTE(cur)                                                                        \
    {                                                                                          \
        const float* Ap = As + (cur) * TS + (32 * wm + li) * A_LD + 4 * hi;                    \
        const float* Bp = Bs + (cur) * TS + (32 * wn + li) * A_LD + 4 * hi;                    \
        _Pragma("unroll") for (int q = 0; q < BK / 8; ++q) {                                   \
            const float4 av = *reinterpret_cast<const float4*>(Ap + 8 * q);                    \
            const float4 bv = *reinterpret_cast<const float4*>(Bp + 8 * q);                    \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);              \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);              \
        }                                                                                      \
    }
    const int nch = a.Bs / BK;
    pa0 = pa1 = pb0 = pb1 = qa0 = qa1 = qb0 = qb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    WB_LOAD(p, 0);
    if (nch > 1) WB_LOAD(q, 1);
    WB_STORE(p, 0);
    __syncthreads();
    for (int c = 0; c < nch; c += 2) {
        if (c + 2 < nch) WB_LOAD(p, c + 2);
        WB_COMPUTE(0);
        if (c + 1 < nch) WB_STORE(q, 1);
        __syncthreads();
        if (c + 1 < nch) {
            if (c + 3 < nch) WB_LOAD(q, c + 3);
            WB_COMPUTE(1);
            if (c + 2 < nch) WB_STORE(p, 0);
            __syncthreads();
        }
    }
#undef WB_COMPUTE
#undef WB_LOAD
#undef WB_STORE
    wg_emit16(h, wg_dst(a, a.gW[i], a.poW[i], slice), a.oW[i],
              ((size_t)l * HID + n0 + 32 * wm) * HID + k0 + 32 * wn + li, HID, hi, acc,
              wg_planes_wh(EMIT ? a.whp : nullptr, i - 1, a.L, l, n0 / 32 + wm, k0 + 32 * wn, li));
    if (k0 == 0) {
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            rs0 += __shfl_xor(rs0, off, 64);
            rs1 += __shfl_xor(rs1, off, 64);
        }
        if (s_c4 == 0) {
            const size_t gb = (size_t)l * HID + n0 + s_row;
            const WgDst db = wg_dst(a, a.gb[i], a.pob[i], slice);
            wg_emit1(h, db, a.ob[i], gb, rs0);
            wg_emit1(h, db, a.ob[i], gb + 32, rs1);
        }
    }
}

__device__ __forceinline__ void wgrad_tile_C(const WgradArgs& a, const NsvdHyper& h, float* lds, int unit, int slice) {
    const int l = unit >> 2, part = unit & 3;  // 4 workgroups per head: 32 of the 128 rows each
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int nh = a.nlayers - 1;
    float* red = lds;            // [8]
    float* dbl = lds + 16;       // [Bs] dbase of this head (Bs <= 2 * 128 * 36 - 16 floats, checked on the host)
    const size_t row0 = (size_t)l * a.B + (size_t)slice * a.Bs;  // this slice's rows of the head's (B) vectors
    float sb = 0.f, ss = 0.f;
    for (int b = tid; b < a.Bs; b += 256) {
        const float v = a.dbase[row0 + b];
        dbl[b] = v;
        sb += v;
        if (a.dfsc) ss += a.dfsc[row0 + b];
    }
    sb = nsvd_wave_sum(sb);
    ss = nsvd_wave_sum(ss);
    if (lane == 0) {
        red[w] = sb;
        red[4 + w] = ss;
    }
    __syncthreads();
    if (tid == 0 && part == 0) {
        wg_emit1(h, wg_dst(a, a.gb[nh], a.pob[nh], slice), a.ob[nh], l, (red[0] + red[1]) + (red[2] + red[3]));
        if (a.dfsc)
            wg_emit1(h, wg_dst(a, a.gscales, a.poscales, slice), a.oscales, l,
                     (red[4] + red[5]) + (red[6] + red[7]));
    }
    if (a.loss_part && unit == 0 && slice == 0 && w == 0) {
        // the step's loss from the chain kernel's partial sums (direct mode), in a fixed order: lane q adds the
        // entries q, q + 64, ...; then the butterfly
        const int nop = a.loss_L * (a.B / BS);
        float op = 0.f, mt = 0.f;
        for (int q = lane; q < nop; q += 64) op += a.loss_part[q];
        for (int q = lane; q < a.loss_L; q += 64) mt += a.loss_part[nop + q];
        op = -2.f * (nsvd_wave_sum(op) / (float)a.B);
        mt = nsvd_wave_sum(mt);
        if (lane == 0) {
            a.loss[0] = op + mt;
            a.loss[1] = op;
            a.loss[2] = mt;
        }
    }
    if (tid == 0 && unit == 0 && slice == 0) {
        // last kernel of the step (no second pass): nothing in this launch reads `step`
        if (a.state && a.S == 1) a.state->step += 1;
    }
    // dW_last[n] = sum_b dbase[b] softplus(z[n][b]): each wave owns 32 rows and walks them 8 at a time so
    // that 8 independent 16-B loads are in flight per lane (a row-at-a-time loop is pure L2 latency)
    {
        const int n0 = 32 * part + 8 * w;  // this wave's 8 rows
        float s[8], s0[8];  // s0: db_0[n] = sum_b dz_0[n][b], the first layer's bias gradient, for the same rows
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = s0[j] = 0.f;
        const float* zrow = a.zsave[nh - 1] + (size_t)slice * a.Bs;
        const float* drow = a.dz[0] + (size_t)slice * a.Bs;
        for (int b = 4 * lane; b < a.Bs; b += 256) {
            float4 z[8], g0[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                z[j] = *reinterpret_cast<const float4*>(zrow + ((size_t)l * HID + n0 + j) * a.B + b);
                g0[j] = *reinterpret_cast<const float4*>(drow + ((size_t)l * HID + n0 + j) * a.B + b);
            }
            const float4 d = *reinterpret_cast<const float4*>(dbl + b);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s[j] = fmaf(d.x, z[j].x, s[j]);
                s[j] = fmaf(d.y, z[j].y, s[j]);
                s[j] = fmaf(d.z, z[j].z, s[j]);
                s[j] = fmaf(d.w, z[j].w, s[j]);
                s0[j] += (g0[j].x + g0[j].y) + (g0[j].z + g0[j].w);
            }
        }
        // the 16 results go out from 16 lanes at once (lane j: dW_last row j, lane 8 + j: db_0 row j): one round trip
        // of optimiser state instead of 16 dependent ones (these workgroups then end at 35 us instead of 43)
        float mine = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float t = nsvd_wave_sum(s[j]);
            const float t0 = nsvd_wave_sum(s0[j]);
            if (lane == j) mine = t;
            if (lane == 8 + j) mine = t0;
        }
        if (lane < 8)
            wg_emit1(h, wg_dst(a, a.gW[nh], a.poW[nh], slice), a.oW[nh], (size_t)l * HID + n0 + lane, mine);
        else if (lane < 16)
            wg_emit1(h, wg_dst(a, a.gb[0], a.pob[0], slice), a.ob[0], (size_t)l * HID + n0 + lane - 8, mine);
    }
}

// NJ: the dW_0 tile shape (2: 128 x 128, 1: 128 x 64 = WgradArgs::tw 64); two kernels rather than one with both shapes
// inside - with all six tile variants in one function the register allocator spills in the 128 x 128 ones
// EMIT: bf16x3 steps - the optimiser epilogues of W_0 and W_1 .. also write the bf16 planes of the updated values
template <int NJ, bool EMIT = false>
__global__ void __launch_bounds__(256, 2) pmlp_fused_wgrad_kernel(WgradArgs a) {
    __shared__ __attribute__((aligned(16))) float smem_wg[4 * HID * A_LD];  // 72 KB: two blocks per CU
    static_assert(4 * HID * A_LD >= NSVD_TNT_FLOATS, "tile_nt buffers must fit the weight-gradient LDS");
    static_assert(4 * HID * A_LD >= T128D_LDS_FLOATS, "the dW_0 tile's DMA ring must fit the weight-gradient LDS");
    float* As = smem_wg;
    float* Bs = smem_wg + 2 * HID * A_LD;
    // the step's optimiser scalars: launch arguments, or read from the device-resident schedule (uniform loads)
    NsvdHyper h = a.h;
    if (a.hstate) h = *nsvd_state_hyper(a.hstate);
    // grid = S x (nA | nB | 4 L) blocks, kind-major so that the long dW_0 tiles are dispatched first
    int bid = blockIdx.x + a.bid0;
    if (bid < a.nA * a.S) {
        const int slice = bid / a.nA;
        bid -= slice * a.nA;
        // the tiles of an XCD (block id mod 8) share heads and feature tiles: pmlp_common.h
        int tl, tk;
        xcd_block_map(bid, a.hx, a.L, a.nA / a.L, tl, tk);
        const int unit = tl * (a.nA / a.L) + tk;
        // the optimiser state rides under the K loop when the step is fused, nothing else is written and the
        // loop has the four peeled chunks the prefetch hangs on
        const bool pf = a.S == 1 && a.opt && !a.gW[0] && a.Bs >= 4 * BK;
        if (!pf) wgrad_tile_A<0, NJ, EMIT>(a, h, smem_wg, unit, slice);
        else if (a.oW[0].ema) wgrad_tile_A<2, NJ, EMIT>(a, h, smem_wg, unit, slice);
        else wgrad_tile_A<1, NJ, EMIT>(a, h, smem_wg, unit, slice);
        return;
    }
    bid -= a.nA * a.S;
    if (bid < a.nB * a.S) {
        WG_STAMP(0, 2ull);
        WG_STAMP(1, wall_clock64());
        if (a.Bs % NSVD_TNT_KC == 0) wgrad_tile_B64<EMIT>(a, h, smem_wg, bid % a.nB, bid / a.nB);
        else wgrad_tile_B<EMIT>(a, h, As, Bs, bid % a.nB, bid / a.nB);
        WG_STAMP(6, wall_clock64());
        return;
    }
    bid -= a.nB * a.S;
    WG_STAMP(0, 3ull);
    WG_STAMP(1, wall_clock64());
    wgrad_tile_C(a, h, As, bid % (4 * a.L), bid / (4 * a.L));
    WG_STAMP(6, wall_clock64());
}

// Split-K second pass: gradient = sum of the S partial slices (in slice order), then stored and / or applied
// (RMSprop + EMA) exactly as the S = 1 epilogue does. One launch over all tensors of the model.
struct ReduceArgs {
    const float* part;
    size_t part_stride;
    int S, ntensors, opt;
    NsvdHyper h;
    size_t off[2 * NSVD_MAX_LAYERS + 1], n[2 * NSVD_MAX_LAYERS + 1];  // slice offset / element count (multiples of 4)
    float* g[2 * NSVD_MAX_LAYERS + 1];
    NsvdOptPtrs o[2 * NSVD_MAX_LAYERS + 1];
    size_t total4;  // float4 groups over all tensors
    nsvd_step_state* state;  // device-resident schedule (or null): h from state->cur, step incremented here
};

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(ReduceArgs a) {
    NsvdHyper h = a.h;
    if (a.state) h = *nsvd_state_hyper(a.state);
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < a.total4; q += (size_t)gridDim.x * 256) {
        // tensors are laid out back to back (padded to 4 floats) inside a slice: find the one holding group q
        int t = 0;
        size_t e = q * 4;
        while (t + 1 < a.ntensors && e >= a.off[t + 1]) ++t;
        e -= a.off[t];
        if (e >= a.n[t]) continue;  // alignment padding between tensors
        const float* src = a.part + a.off[t] + e;
        float4 gsum = *reinterpret_cast<const float4*>(src);
        for (int sl = 1; sl < a.S; ++sl) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)sl * a.part_stride);
            gsum.x += v.x; gsum.y += v.y; gsum.z += v.z; gsum.w += v.w;
        }
        const int cnt = a.n[t] - e < 4 ? (int)(a.n[t] - e) : 4;
        float gv[4] = {gsum.x, gsum.y, gsum.z, gsum.w};
        if (cnt == 4 && !a.g[t] && a.opt) {
            // the common case as 16-byte accesses (one group of four per thread: 4-byte accesses 16 bytes apart use a
            // quarter of every line per instruction)
            const NsvdOptPtrs& o = a.o[t];
            if ((((uintptr_t)(o.p + e) | (uintptr_t)(o.sq + e) | (uintptr_t)(o.ema ? o.ema + e : o.p + e)) & 15) == 0) {
                float4 pv = *reinterpret_cast<const float4*>(o.p + e), sv = *reinterpret_cast<const float4*>(o.sq + e);
                float4 ev = o.ema ? *reinterpret_cast<const float4*>(o.ema + e) : make_float4(0.f, 0.f, 0.f, 0.f);
                nsvd_rmsprop_upd(pv.x, gv[0], sv.x, ev.x, o.ema != nullptr, h);
                nsvd_rmsprop_upd(pv.y, gv[1], sv.y, ev.y, o.ema != nullptr, h);
                nsvd_rmsprop_upd(pv.z, gv[2], sv.z, ev.z, o.ema != nullptr, h);
                nsvd_rmsprop_upd(pv.w, gv[3], sv.w, ev.w, o.ema != nullptr, h);
                *reinterpret_cast<float4*>(o.p + e) = pv;
                *reinterpret_cast<float4*>(o.sq + e) = sv;
                if (o.ema) *reinterpret_cast<float4*>(o.ema + e) = ev;
                continue;
            }
        }
        for (int c = 0; c < cnt; ++c) {
            if (a.g[t]) a.g[t][e + c] = gv[c];
            if (a.opt) {
                const NsvdOptPtrs& o = a.o[t];
                float pv = o.p[e + c], sv = o.sq[e + c], ev = o.ema ? o.ema[e + c] : 0.f;
                nsvd_rmsprop_upd(pv, gv[c], sv, ev, o.ema != nullptr, h);
                o.p[e + c] = pv;
                o.sq[e + c] = sv;
                if (o.ema) o.ema[e + c] = ev;
            }
        }
    }
    if (a.state && blockIdx.x == 0 && threadIdx.x == 0) a.state->step += 1;  // last kernel of the step
}
#undef WG_LD
#undef WG_ST

}  // namespace


// Head window [l0, l0 + Lc) of the model's d.L heads: the heads of ParallelMLP share nothing but the input, so the
// backward of a window is the same two launches on pointers that start at head l0 (sample-sharded runs cut the
// backward into windows to start exchanging the first window's gradients while the next one is computed).
// win: one of SEVERAL head windows of ONE fused step (nsvd_operator_backward_evd_step_window). not_last: this window
// neither advances the device-resident schedule nor adds up the loss; ev_after_chain: recorded between the two launches
// (the next window's chain, on another stream, waits for it: two chains side by side only slow each other).
struct NsvdStepWindow {
    int not_last;
    hipEvent_t ev_after_chain;
};
static int fused_backward_impl(const nsvd_model_desc& dfull, const nsvd_params& p, int B, const float* df,
                               const NsvdEvdIn* evd, const nsvd_params* gp, const NsvdOptStep* opt, void* ws,
                               hipStream_t s, int l0 = 0, int Lc = 0, const NsvdNextBatch* next = nullptr,
                               const NsvdStepWindow* win = nullptr) {
    if (Lc <= 0) Lc = dfull.L;
    if (l0 < 0 || l0 + Lc > dfull.L) return NSVD_EINVAL;
    nsvd_params g;
    memset(&g, 0, sizeof(g));
    if (gp) g = *gp;
    const FusedWs w = carve_fused(dfull, B, ws);
    nsvd_model_desc d = dfull;
    d.L = Lc;
    const int F = 2 * d.m, nh = d.nlayers - 1;
    if (Lc != dfull.L && wgrad_slices(d, B) != 1) return NSVD_EUNSUPPORTED;  // split-K partials are laid out per model
    // element offsets of head l0 inside the per-head arrays
    auto offW = [&](int i) { return (size_t)l0 * d.dims[i] * (i == 0 ? (size_t)F : (size_t)d.dims[i - 1]); };
    auto offb = [&](int i) { return (size_t)l0 * d.dims[i]; };
    const size_t offz = (size_t)l0 * HID * B, offv = (size_t)l0 * B;

    ChainArgs a;
    memset(&a, 0, sizeof(a));
    a.df = df;
    if (evd) a.evd = *evd;
    a.evd.l_off += l0;
    a.jac = w.jac;
    a.dsc = d.has_exp_mask ? w.dsc : nullptr;
    a.dbase = w.dbase + offv;
    a.dfsc = d.has_exp_mask ? w.dfsc + offv : nullptr;
    for (int i = 0; i < d.nlayers; ++i) {
        a.W[i] = p.W[i] + offW(i);
        a.zsave[i] = (i < nh) ? w.zsave[i] + offz : nullptr;
        a.dz[i] = (i < nh) ? w.dz[i] + offz : nullptr;
    }
    a.nlayers = d.nlayers; a.B = B; a.L = d.L;
    a.ldl = dfull.L; a.l0 = l0;
    const int chain_only = (B / BS) * d.L;
    a.chain_blocks = chain_only;
    int chain_grid = chain_only;
    nsvd_step_state* state = opt ? opt->state : nullptr;
    a.state = state;
    // the loss of the step in direct mode (no moment kernel): only when this launch sees every head of the loss
    const bool direct = evd && !evd->moments && !evd->part;
    const bool step_loss = direct && evd->loss && B / BS <= 32 && (Lc == dfull.L || win) && evd->l_off == 0 &&
                           evd->Lg == dfull.L && !df;
    a.loss_part = step_loss ? w.loss_part : nullptr;
    a.loss_L = dfull.L;
    if (next) {  // the next batch's sampling + features as guest workgroups (same arguments as nsvd_fused_features)
        const FusedWs wn = carve_fused(dfull, B, next->ws);
        memset(&a.feat, 0, sizeof(a.feat));
        a.feat.smp = next->smp;
        a.feat.x = next->x; a.feat.xout = next->x;
        a.feat.fB = p.fourier_B; a.feat.phi = wn.phi; a.feat.phiTc = wn.phiTc; a.feat.sctab = wn.sctab;
        a.feat.B = B; a.feat.m = d.m; a.feat.D = d.D; a.feat.eps = next->eps;
        if (opt && opt->state) a.feat.smp.offset_add = (const unsigned long long*)&opt->state->step;
        a.feat_nx = (d.m + nsvd_feat::FJ - 1) / nsvd_feat::FJ;
        a.feat_blocks = a.feat_nx * ((B + nsvd_feat::FB - 1) / nsvd_feat::FB);
        chain_grid += a.feat_blocks;
    }
    if (state) chain_grid += 1;  // the schedule block
    // the streaming form (pmlp_stream_bwd.h) where the shape allows: no dz_i through HBM, the slices' partial gradients
    // through the split-K second pass below
    int SS = 0;
    if (nh == 2 && F == HID && !next && !win && !state && Lc == dfull.L && l0 == 0 && w.gpart &&
        (df || (evd && (evd->moments || evd->part) && evd->Lg % 4 == 0 && evd->Lg <= 128))) {
        static const char* e = getenv("NSVD_STREAM_BWD");
        if (!(e && e[0] == '0')) SS = stream_bwd_slices(d, B);
    }
    const PartLayout pl = part_layout(d);
    if (SS) {
        StreamArgs sa;
        memset(&sa, 0, sizeof(sa));
        sa.df = df; sa.jac = a.jac; sa.dsc = a.dsc;
        sa.W1 = a.W[1]; sa.Wl = a.W[2]; sa.a0 = a.zsave[0]; sa.a1 = a.zsave[1]; sa.phiTc = w.phiTc;
        sa.B = B; sa.L = d.L; sa.ldl = a.ldl; sa.l0 = a.l0; sa.S = SS; sa.Bs = B / SS;
        sa.evd = a.evd;
        sa.part = w.gpart; sa.part_stride = pl.stride;
        for (int i = 0; i < 3; ++i) {
            sa.poW[i] = pl.oW[i];
            sa.pob[i] = pl.ob[i];
        }
        sa.poscales = pl.oscales;
        {
            static const char* e2 = getenv("NSVD_STREAM_DBG");
            sa.dbg = e2 ? atoi(e2) : 0;
        }
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)pmlp_stream_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)SB_LDS_BYTES);
            if (e != hipSuccess) return -(int)e;
            attr_set = true;
        }
        static const char* e3 = getenv("NSVD_STREAM_BWD");
        if (e3 && e3[0] == '1') {  // the one-group form (pmlp_stream_bwd.h), kept for A/B measurements
            hipLaunchKernelGGL(pmlp_stream_bwd_kernel, dim3(d.L * SS), dim3(256), SB_LDS_BYTES, s, sa);
        } else {
            static bool attr2_set = false;
            if (!attr2_set) {
                hipError_t er = hipFuncSetAttribute((const void*)pmlp_stream_bwd2_kernel,
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)SB2_LDS_BYTES);
                if (er != hipSuccess) return -(int)er;
                attr2_set = true;
            }
            hipLaunchKernelGGL(pmlp_stream_bwd2_kernel, dim3(d.L * SS), dim3(512), SB2_LDS_BYTES, s, sa);
        }
    } else if ((nh == 2 || nh == 3) && chain_only <= 128)
        hipLaunchKernelGGL(pmlp_fused_bwd_chain_kernel<true>, dim3(chain_grid), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(pmlp_fused_bwd_chain_kernel<false>, dim3(chain_grid), dim3(256), 0, s, a);
    NSVD_CHECK_LAUNCH();
    if (win && win->ev_after_chain) {
        hipError_t e = hipEventRecord(win->ev_after_chain, s);
        if (e != hipSuccess) return -(int)e;
    }

    WgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    for (int i = 0; i < d.nlayers; ++i) {
        wa.dz[i] = (i < nh) ? w.dz[i] + offz : nullptr;
        wa.zsave[i] = (i < nh) ? w.zsave[i] + offz : nullptr;
        wa.gW[i] = g.W[i] ? g.W[i] + offW(i) : nullptr;
        wa.gb[i] = g.b[i] ? g.b[i] + offb(i) : nullptr;
    }
    wa.phiTc = w.phiTc;
    wa.dbase = w.dbase + offv;
    wa.dfsc = d.has_exp_mask ? w.dfsc + offv : nullptr;
    wa.gscales = (d.has_exp_mask && g.scales) ? g.scales + l0 : nullptr;
    wa.nlayers = d.nlayers; wa.B = B; wa.L = d.L; wa.F = F;
    if (opt) {
        wa.opt = 1;
        wa.h = opt->h;
        for (int i = 0; i < d.nlayers; ++i) {
            wa.oW[i] = NsvdOptPtrs{p.W[i] + offW(i), opt->sq.W[i] + offW(i), opt->ema ? opt->ema->W[i] + offW(i) : nullptr};
            wa.ob[i] = NsvdOptPtrs{p.b[i] + offb(i), opt->sq.b[i] + offb(i), opt->ema ? opt->ema->b[i] + offb(i) : nullptr};
        }
        if (d.has_exp_mask)
            wa.oscales = NsvdOptPtrs{p.scales + l0, opt->sq.scales + l0, opt->ema ? opt->ema->scales + l0 : nullptr};
    }
    wa.state = (win && win->not_last) ? nullptr : state;  // (the optimiser reads the schedule through wa.hstate)
    wa.hstate = state;
    if (step_loss && !(win && win->not_last)) {
        wa.loss_part = w.loss_part;
        wa.loss = evd->loss;
        wa.loss_L = dfull.L;
    }
    wa.nA = (F / HID) * d.L;
    wa.S = wgrad_slices(d, B);
    wa.Bs = B / wa.S;
    wa.nB = 4 * (nh - 1) * d.L;
    if (SS) wa.S = SS;
    if (wa.S > 1) {
        wa.part = w.gpart;
        wa.part_stride = pl.stride;
        for (int i = 0; i < d.nlayers; ++i) {
            wa.poW[i] = pl.oW[i];
            wa.pob[i] = pl.ob[i];
        }
        wa.poscales = pl.oscales;
    }
    wa.tw = 128;
    if (wgrad_tile_width(wa.nA, wa.S) == 64) {  // 128 x 64 tiles: twice as many, half as long (pmlp_common.h)
        wa.tw = 64;
        wa.nA *= 2;
    }
    wa.hx = pick_xcd_remap_wgrad(d.L, wa.nA / d.L, wa.tw);
    // One launch: the dW_0 tiles go one per CU first, the small dW_i / db / last-layer workgroups
    // then co-reside with them (measured: 55 us together vs 42 + 23 us as two launches).
    wa.bid0 = 0;
    // bf16x3 steps: the planes of the updated weights go where the NEXT forward reads them (the other workspace set
    // when the next batch's features ride along, this one otherwise); whole model, fused optimiser, no split-K only
    // (never with the streaming form: it skips the kernel that writes the planes)
    const bool emit = opt && opt->emit_planes && !SS && wa.S == 1 && Lc == dfull.L && l0 == 0;
    if (emit) {
        const FusedWs wp = carve_fused(dfull, B, next ? next->ws : ws);
        wa.w0p = reinterpret_cast<unsigned short*>(wp.w0p);
        wa.whp = reinterpret_cast<unsigned short*>(wp.whp);
    }
    const dim3 wgrid(wa.S * (wa.nA + wa.nB + 4 * d.L));
    if (SS) {
        // (the streaming kernel has written every slice)
    } else if (wa.tw == 64) {
        if (emit) hipLaunchKernelGGL((pmlp_fused_wgrad_kernel<1, true>), wgrid, dim3(256), 0, s, wa);
        else hipLaunchKernelGGL(pmlp_fused_wgrad_kernel<1>, wgrid, dim3(256), 0, s, wa);
    } else {
        if (emit) hipLaunchKernelGGL((pmlp_fused_wgrad_kernel<2, true>), wgrid, dim3(256), 0, s, wa);
        else hipLaunchKernelGGL(pmlp_fused_wgrad_kernel<2>, wgrid, dim3(256), 0, s, wa);
    }
    NSVD_CHECK_LAUNCH();
    if (wa.S == 1 && !SS) return 0;
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.part = w.gpart;
    ra.part_stride = pl.stride;
    ra.S = wa.S;
    ra.opt = wa.opt;
    ra.h = wa.h;
    ra.state = state;
    int t = 0;
    for (int i = 0; i < d.nlayers; ++i, ++t) {
        ra.off[t] = pl.oW[i]; ra.n[t] = pl.nW[i]; ra.g[t] = g.W[i]; ra.o[t] = wa.oW[i];
    }
    for (int i = 0; i < d.nlayers; ++i, ++t) {
        ra.off[t] = pl.ob[i]; ra.n[t] = pl.nb[i]; ra.g[t] = g.b[i]; ra.o[t] = wa.ob[i];
    }
    if (d.has_exp_mask) {
        ra.off[t] = pl.oscales; ra.n[t] = pl.nscales; ra.g[t] = g.scales; ra.o[t] = wa.oscales;
        ++t;
    }
    ra.ntensors = t;
    ra.total4 = (pl.oscales + (pl.nscales + 3) / 4 * 4) / 4;
    size_t blocks = (ra.total4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ra);
    NSVD_CHECK_LAUNCH();
    return 0;
}

// developer diagnostic (not in include/nsvd.h): the streaming backward's per-region cycle counts (NSVD_STREAM_DBG = 4)
extern "C" int nsvd_debug_stream_stamps(unsigned long long* host) {
    return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sb_stamps), 8 * sizeof(unsigned long long));
}

#ifdef NSVD_WG_STAMPS
extern "C" int nsvd_debug_wgrad_stamps(unsigned long long* host, size_t n) {
    return -(int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg_stamps), n * sizeof(unsigned long long));
}
#endif

int nsvd_fused_backward(const nsvd_model_desc& d, const nsvd_params& p, const nsvd_problem& prob, const float* x,
                        int B, const float* df, const nsvd_params& g, void* ws, hipStream_t s) {
    (void)prob;
    (void)x;
    return fused_backward_impl(d, p, B, df, nullptr, &g, nullptr, ws, s);
}

int nsvd_fused_backward_evd(const nsvd_model_desc& d, const nsvd_params& p, int B, const NsvdEvdIn& evd,
                            const nsvd_params* g, const NsvdOptStep* opt, void* ws, hipStream_t s, int l_begin,
                            int l_count, const NsvdNextBatch* next, int window_of_step, int not_last,
                            void* ev_after_chain) {
    if (!g && !opt) return NSVD_EINVAL;
    if (next && (d.D < 1 || d.D > 3 || !p.fourier_B || !next->ws || !next->x)) return NSVD_EINVAL;
    NsvdStepWindow win;
    win.not_last = not_last;
    win.ev_after_chain = (hipEvent_t)ev_after_chain;
    return fused_backward_impl(d, p, B, nullptr, &evd, g, opt, ws, s, l_begin, l_count, next,
                               window_of_step ? &win : nullptr);
}

int nsvd_fused_wgrad_slices(const nsvd_model_desc& d, int B) { return wgrad_slices(d, B); }
int nsvd_fused_stream_bwd_slices(const nsvd_model_desc& d, int B) { return stream_bwd_slices(d, B); }

bool nsvd_fused_backward_window_ok(const nsvd_model_desc& d, int B, int l_count) {
    if (l_count <= 0 || l_count > d.L) return false;
    if (l_count == d.L) return true;
    nsvd_model_desc dw = d;
    dw.L = l_count;
    return wgrad_slices(dw, B) == 1;
}
