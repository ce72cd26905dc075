// Persistent LayerNorm + LSTM recurrence kernel (the dominant kernel: 77 % of the path's FLOPs,
// SURVEY.md §8a rows a8/a9/a11).
//
// One workgroup (4 waves, one per SIMD) owns NS = 16*MT sequences of one direction for ALL time steps:
//   * [W_ih | W_hh]^T (128 x 256 fp32) lives in VGPRs for the whole kernel as MFMA B-operand fragments:
//     wave w owns hidden units 16w..16w+15 of all four gates (i,f,g,o), i.e. 4 gate tiles x 32 k-steps =
//     128 registers per lane, so the cell update is lane-local (all four gates of a (sequence, unit) pair
//     land in the same lane of the v_mfma_f32_16x16x4_f32 accumulators);
//   * per step the A operand [NS x 128] = [LayerNorm(x_t) | h_{t-1}] is staged in LDS (double buffered,
//     one barrier per step); k is permuted so each lane reads 32 contiguous floats (ds_read_b128);
//   * exact fp32 MFMA (bitwise an fmaf chain) keeps the 625-step recurrences inside the 1e-3 budget;
//   * x_{t+1} is fetched and normalised while step t computes; h_t is written back coalesced from LDS.
#include <type_traits>

#include "lh_common.h"

namespace lh {

constexpr int LS_PAD = 36;   // LDS row of a 32-float k-chunk (+4 floats: conflict-free ds_read_b128, 16 B aligned)

template <int MT>
__global__ void __launch_bounds__(256) k_ln_lstm(const float* __restrict__ x, const float* __restrict__ lnw,
                                                 const float* __restrict__ lnb, const float* __restrict__ w_pk,
                                                 const float* __restrict__ b_sum, const float* __restrict__ h0,
                                                 const float* __restrict__ c0, float* __restrict__ hN,
                                                 float* __restrict__ cN, float* __restrict__ h_out, int nseq,
                                                 int nstep, int sdiv, int so, int si, int ps, int ldh) {
    constexpr int NS = 16 * MT;
    __shared__ __attribute__((aligned(16))) float abuf[2 * 4 * NS * LS_PAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * NS;
    const int g4 = lane >> 4, l15 = lane & 15;

    auto A = [&](int buf, int chunk, int row, int j) -> float* {
        return &abuf[((buf * 4 + chunk) * NS + row) * LS_PAD + j];
    };
    auto row_of = [&](int s, int p) -> long { return (long)(s / sdiv) * so + (long)(s % sdiv) * si + (long)p * ps; };
    auto step_pos = [&](int it) -> int { return dir ? (nstep - 1 - it) : it; };

    // ---- resident weights: B-operand fragments of this wave's 64 gate columns
    float wreg[4][32];
    {
        const float* wp = w_pk + ((long)(dir * 4 + wave) * 4) * 32 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < 32; ++ks) wreg[g][ks] = wp[(g * 32 + ks) * 64];
    }
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = b_sum[dir * 256 + g * 64 + wave * 16 + l15];

    // ---- per-thread role in the row-wise phases: 16 lanes per 64-float row, float4 each
    const int q = tid & 15;                       // float4 index within the row
    const float4 gw = *reinterpret_cast<const float4*>(&lnw[q * 4]);
    const float4 gb = *reinterpret_cast<const float4*>(&lnb[q * 4]);

    auto load_x = [&](int it, float4 (&xr)[MT]) {
        const int p = step_pos(it);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = min(s0 + rl, nseq - 1);
            xr[i] = *reinterpret_cast<const float4*>(&x[row_of(s, p) * C + q * 4]);
        }
    };
    auto norm_store_x = [&](int buf, float4 (&xr)[MT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            float4 v = xr[i];
            const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
            const float rstd = rsqrtf(var + LN_EPS);
            float4 y;
            y.x = v.x * rstd * gw.x + gb.x; y.y = v.y * rstd * gw.y + gb.y;
            y.z = v.z * rstd * gw.z + gb.z; y.w = v.w * rstd * gw.w + gb.w;
            *reinterpret_cast<float4*>(A(buf, q >> 3, rl, (q & 7) * 4)) = y;
        }
    };
    auto flush_h = [&](int buf, int it) {          // h of step `it` (in LDS buffer `buf`) -> global, coalesced
        const int p = step_pos(it);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = s0 + rl;
            if (s < nseq) {
                const float4 hv = *reinterpret_cast<const float4*>(A(buf, 2 + (q >> 3), rl, (q & 7) * 4));
                *reinterpret_cast<float4*>(&h_out[row_of(s, p) * ldh + dir * H + q * 4]) = hv;
            }
        }
    };

    // ---- prologue: x of step 0, initial state
    float creg[MT][4];
    {
        float4 xr[MT];
        load_x(0, xr);
        norm_store_x(0, xr);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = min(s0 + rl, nseq - 1);
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h0) hv = *reinterpret_cast<const float4*>(&h0[(long)s * H + q * 4]);
            *reinterpret_cast<float4*>(A(0, 2 + (q >> 3), rl, (q & 7) * 4)) = hv;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = min(s0 + m * 16 + g4 * 4 + r, nseq - 1);
                creg[m][r] = c0 ? c0[(long)s * H + wave * 16 + l15] : 0.0f;
            }
    }
    __syncthreads();

    for (int it = 0; it < nstep; ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        float4 xr[MT];
        const bool more = (it + 1 < nstep);
        if (more) load_x(it + 1, xr);             // in flight during the MFMA phase
        if (it > 0) flush_h(cur, it - 1);

        // gates = [x_t | h_{t-1}] * [W_ih | W_hh]^T + b   (K = 128 as 32 k-steps of 4)
        f32x4 acc[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[m][g] = f32x4{bias[g], bias[g], bias[g], bias[g]};
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float* arow = A(cur, g4, m * 16 + l15, 0);
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        acc[m][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wreg[g][qq * 4 + j], acc[m][g], 0, 0, 0);
            }
        }

        // cell update, lane-local: accumulator reg r <-> sequence row m*16 + g4*4 + r, unit wave*16 + l15
        const int unit = wave * 16 + l15;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = sigmoid_f(acc[m][0][r]);
                const float fg = sigmoid_f(acc[m][1][r]);
                const float gg = tanh_f(acc[m][2][r]);
                const float og = sigmoid_f(acc[m][3][r]);
                const float cc = fg * creg[m][r] + ig * gg;
                creg[m][r] = cc;
                *A(nxt, 2 + (unit >> 5), m * 16 + g4 * 4 + r, unit & 31) = og * tanh_f(cc);
            }
        if (more) norm_store_x(nxt, xr);
        __syncthreads();
    }

    // ---- epilogue: last hidden state to the sequence output, final (h, c) to the carried state
    const int last = nstep & 1;
    flush_h(last, nstep - 1);
    if (hN) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = s0 + rl;
            if (s < nseq)
                *reinterpret_cast<float4*>(&hN[(long)s * H + q * 4]) =
                    *reinterpret_cast<const float4*>(A(last, 2 + (q >> 3), rl, (q & 7) * 4));
        }
    }
    if (cN) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = s0 + m * 16 + g4 * 4 + r;
                if (s < nseq) cN[(long)s * H + wave * 16 + l15] = creg[m][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------------
// Split-precision variant ("f16x3"): every fp32 operand v is carried as two fp16 numbers
//     hi = fp16(v),  lo = fp16(v - hi)                              (v - hi is exact in fp32)
// and a product a*b is evaluated as  hi_a*hi_b + hi_a*lo_b + lo_a*hi_b  on v_mfma_f32_16x16x32_f16, all three into
// ONE fp32 accumulator (fp16 products are exact in it; the dropped lo*lo term is 2^-22 relative).  `lo` is NOT
// rescaled: the matrix core takes fp16 subnormals at full value (scripts/ubench/gen_issue_model.py denorm_test,
// profiles/r02a_ubench_issue_model.txt), so lo keeps 11 bits while |v| >= 2^-3 and an absolute 2^-25 below that —
// for the operands of these kernels (LayerNorm outputs, h in (-1, 1), weights of O(0.1)) that is the fp32 noise level.
// Not rescaling saves the second accumulator set (16 VGPRs), the 2^-11 recombination (16 + 4 VALU per step) and one
// multiply per split element.  The measured end-to-end error stays at the 1e-6 level against the 1e-3 budget, while
// the three fp16 MFMAs (K=32 each) cost ~1/5 of the fp32 MFMA (K=4) they replace.  Same persistent structure:
// weights resident in VGPRs (now as hi/lo fp16 fragments, same 128 registers), A = [LN(x_t) | h_{t-1}] staged
// in LDS as fp16 hi/lo rows, one barrier per step.
// ------------------------------------------------------------------------------------------------------
#if defined(LH_PROBE_TRACE)        // timing probe build only: per-step s_memtime stamps of two workgroups of k_ln_lstm_lin
__device__ unsigned long long lh_trace_buf[2 * 128 * 4];
#endif
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int LH_AP = 144;      // fp16 elements per LDS row: 128 + 16 pad (288 B: conflict-free ds_read_b128)
constexpr int LH_HP = 68;       // fp32 copy of h: 64 + 4 pad

__device__ __forceinline__ void split_f16(float v, _Float16& hi, _Float16& lo) { split_hl(v, hi, lo); }

template <int MT>
__global__ void __launch_bounds__(256, 2) k_ln_lstm_h3(const float* __restrict__ x, const float* __restrict__ lnw,
                                                    const float* __restrict__ lnb, const _Float16* __restrict__ w_pk,
                                                    const float* __restrict__ b_sum, const float* __restrict__ h0,
                                                    const float* __restrict__ c0, float* __restrict__ hN,
                                                    float* __restrict__ cN, float* __restrict__ h_out, int nseq,
                                                    int nstep, int sdiv, int so, int si, int ps, int ldh) {
    constexpr int NS = 16 * MT;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * NS * LH_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * NS * LH_AP];
    __shared__ __attribute__((aligned(16))) float hf[2 * NS * LH_HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    const int dir = blockIdx.y;
    const int s0 = blockIdx.x * NS;
    const int g4 = lane >> 4, l15 = lane & 15;

    auto row_of = [&](int s, int p) -> long { return (long)(s / sdiv) * so + (long)(s % sdiv) * si + (long)p * ps; };
    // clamped: fetches/stores past either end of the sequence hit an in-range row and are harmless
    auto step_pos = [&](int it) -> int { it = min(max(it, 0), nstep - 1); return dir ? (nstep - 1 - it) : it; };

    // resident weights: hi/lo fp16 B fragments, [gate][kstep]; image = [dir][wave][gate][ks][lane][hi8|lo8]
    f16x8 wh[4][4], wl[4][4];
    {
        const _Float16* wp = w_pk + ((long)(dir * 4 + wave) * 16 * 64 + lane) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wh[g][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(g * 4 + ks) * 64 * 16);
                wl[g][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(g * 4 + ks) * 64 * 16 + 8);
            }
    }
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = b_sum[dir * 256 + g * 64 + wave * 16 + l15];

    const int q = tid & 15;
    // the LayerNorm affine is folded into the packed image: W_ih' = W_ih * ln_w, b' = b + W_ih ln_b (weights.py)

    auto load_x = [&](int it, float4 (&xr)[MT]) {
        const int p = step_pos(it);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = min(s0 + rl, nseq - 1);
            xr[i] = *reinterpret_cast<const float4*>(&x[row_of(s, p) * C + q * 4]);
        }
    };
    auto store_split4 = [&](int buf, int rl, int col, float a, float b, float c, float d) {
        f16x4 h4, l4;
        _Float16 th, tl;
        split_f16(a, th, tl); h4[0] = th; l4[0] = tl;
        split_f16(b, th, tl); h4[1] = th; l4[1] = tl;
        split_f16(c, th, tl); h4[2] = th; l4[2] = tl;
        split_f16(d, th, tl); h4[3] = th; l4[3] = tl;
        *reinterpret_cast<f16x4*>(&ahi[(buf * NS + rl) * LH_AP + col]) = h4;
        *reinterpret_cast<f16x4*>(&alo[(buf * NS + rl) * LH_AP + col]) = l4;
    };
    auto norm_store_x = [&](int buf, float4 (&xr)[MT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            float4 v = xr[i];
            const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
            const float rstd = rsqrtf(var + LN_EPS);
            store_split4(buf, rl, q * 4, v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd);
        }
    };
    auto flush_h = [&](int buf, int it) {
        const int p = step_pos(it);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            // tail rows (s >= nseq) are exact replicas of sequence nseq-1 (all their inputs are clamped), so the
            // clamped, unconditional store rewrites identical bytes and the step body stays branch-free
            const int s = min(s0 + rl, nseq - 1);
            *reinterpret_cast<float4*>(&h_out[row_of(s, p) * ldh + dir * H + q * 4]) =
                *reinterpret_cast<const float4*>(&hf[(buf * NS + rl) * LH_HP + q * 4]);
        }
    };

    float creg[MT][4];
    float4 xr[MT];
    {
        load_x(0, xr);
        norm_store_x(0, xr);
        load_x(1, xr);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = min(s0 + rl, nseq - 1);
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h0) hv = *reinterpret_cast<const float4*>(&h0[(long)s * H + q * 4]);
            store_split4(0, rl, C + q * 4, hv.x, hv.y, hv.z, hv.w);
            *reinterpret_cast<float4*>(&hf[rl * LH_HP + q * 4]) = hv;
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = min(s0 + m * 16 + g4 * 4 + r, nseq - 1);
                creg[m][r] = c0 ? c0[(long)s * H + wave * 16 + l15] : 0.0f;
            }
    }
    __syncthreads();

    for (int it = 0; it < nstep; ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        // One straight-line block per step (no data-dependent branches) so the scheduler can slide the row-wise
        // VALU work under the MFMAs: x_{t+1} (fetched a step ago) is normalised into the free buffer first,
        // x_{t+2} is fetched, h_{t-1} is flushed (at t = 0 the initial state goes to the row of step 0, which
        // the real h_0 overwrites one step later from the same thread).
        norm_store_x(nxt, xr);
        load_x(it + 2, xr);
        flush_h(cur, it - 1);

        // Per 16-sequence tile: 48 MFMAs, then the two split accumulators collapse to 16 gate pre-activations so
        // the accumulator registers are free again; with MT = 2 the second tile's MFMAs have no dependence on the
        // first tile's cell update and the scheduler overlaps matrix and VALU work inside the wave.
        f32x4 gate[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[m][g] = f32x4{bias[g], bias[g], bias[g], bias[g]};
            const int ro = (cur * NS + m * 16 + l15) * LH_AP + g4 * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[ro + ks * 32]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[ro + ks * 32]);
#pragma unroll
                for (int g = 0; g < 4; ++g) gate[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[g][ks], gate[m][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) gate[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[g][ks], gate[m][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) gate[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[g][ks], gate[m][g], 0, 0, 0);
            }
        }

        const int unit = wave * 16 + l15;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float hv;
                lstm_cell_pre(gate[m][0][r], gate[m][1][r], gate[m][2][r], gate[m][3][r], creg[m][r], hv);
                const int rl = m * 16 + g4 * 4 + r;
                _Float16 th, tl;
                split_f16(hv, th, tl);
                ahi[(nxt * NS + rl) * LH_AP + C + unit] = th;
                alo[(nxt * NS + rl) * LH_AP + C + unit] = tl;
                hf[(nxt * NS + rl) * LH_HP + unit] = hv;
            }
        __syncthreads();
    }

    const int last = nstep & 1;
    flush_h(last, nstep - 1);
    if (hN) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int rl = (tid + 256 * i) >> 4;
            const int s = s0 + rl;
            if (s < nseq)
                *reinterpret_cast<float4*>(&hN[(long)s * H + q * 4]) =
                    *reinterpret_cast<const float4*>(&hf[(last * NS + rl) * LH_HP + q * 4]);
        }
    }
    if (cN) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = s0 + m * 16 + g4 * 4 + r;
                if (s < nseq) cN[(long)s * H + wave * 16 + l15] = creg[m][r];
            }
    }
}

#if defined(LH_LEGACY)   // A/B lab + emulator builds only (python -m lookoncetohear_amd.build --variant legacy -DLH_LEGACY):
                         // the fused intra kernel of round 2, superseded by k_intra_xp (lh_recur.hip); the product library
                         // contains only kernels a product configuration can launch
// ------------------------------------------------------------------------------------------------------
// Recurrence with the following Linear + residual fused in ("lin" kernels): instead of writing the hidden states
// to HBM for a separate pointwise kernel, every step also multiplies h_{t-1} by the wave's 16 columns of the
// output projection (6 extra MFMAs on the h fragments it has already read), parks the 16 x 64 product in LDS and,
// one step later, the row-wise threads add bias + residual and store the finished 256-byte activation rows.
//   NPASS = 1 (inter path):  out = res + b + W h                                  (tfgridnet_causal.py:534-538)
//   NPASS = 2 (intra path):  the same workgroup runs the forward direction (out = res + b + W[:, :64] h_fwd)
//                            and then the reverse direction (out += W[:, 64:] h_bwd) over its sequences, so
//                            both halves of the bidirectional projection meet in the same rows without any
//                            inter-workgroup hand-off (:505-516).
// HBM traffic per block drops from (LSTM 4A + Linear 4A) to <= 5A intra and from (2A + 3A) to 2A inter.
// ------------------------------------------------------------------------------------------------------
template <int MT>
__global__ void __launch_bounds__(256, 2) k_ln_lstm_lin(const float* __restrict__ x, const _Float16* __restrict__ w_pk,
                                                        const float* __restrict__ b_sum, const _Float16* __restrict__ wlin_pk,
                                                        const float* __restrict__ blin, const float* __restrict__ h0,
                                                        const float* __restrict__ c0, float* __restrict__ hN,
                                                        float* __restrict__ cN, float* out, int nseq,
                                                        int nstep, int sdiv, int so, int si, int ps, int dir,
                                                        int accumulate, int dephase) {
    constexpr int NS = 16 * MT;
    constexpr int LSP = C + 4;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * NS * LH_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * NS * LH_AP];
    __shared__ __attribute__((aligned(16))) float ls[2 * NS * LSP];
#if defined(LH_PROBE_TRACE)
    __shared__ unsigned long long tr[128 * 4];
    const int tr_slot = blockIdx.x == 7 ? 0 : (blockIdx.x == gridDim.x - 9 ? 1 : -1);
    const bool tr_on = tr_slot >= 0 && threadIdx.x == 0 && dir == 0;
#define LH_STAMP(k) do { if (tr_on) tr[(it & 127) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LH_STAMP(k) do { } while (0)
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    const int s0 = blockIdx.x * NS;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int q = tid & 15;
    const int unit = wave * 16 + l15;
    // The two workgroups sharing a CU start together and, with fair issue arbitration, stay phase-locked: both in the
    // MFMA phase, then both in the activation (VALU) phase, so the matrix and vector pipes never overlap.  Giving the
    // wave in the odd hardware slot a higher issue priority breaks the symmetry: it wins the MFMA pipe, reaches its
    // VALU phase first, and from then on one workgroup's MFMAs run under the other's activations.
    if (dephase) {
        const unsigned hw_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID[3:0] = wave slot
        if (hw_id & 1) __builtin_amdgcn_s_setprio(2);
    }

    // Addressing.  row(s, p) = (s / sdiv) * so + (s % sdiv) * si + p * ps  (256-byte activation rows).  Per thread only
    // the sequence part varies and it is loop-invariant; the step part p * ps is wave-uniform.  So every global access
    // of the step loop is  (uniform 64-bit base of the workgroup's first row, advanced per step on the scalar unit) +
    // (a 32-bit per-thread byte offset computed once): no vector address arithmetic inside the loop.
    auto row_of0 = [&](int s) -> long { return (long)(s / sdiv) * so + (long)(s % sdiv) * si; };
    const long wg_row0 = row_of0(min(s0, nseq - 1));
    unsigned voff[MT];                              // bytes from the workgroup's first row to this thread's float4
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int s = min(s0 + ((tid + 256 * i) >> 4), nseq - 1);      // tail rows replicate sequence nseq-1
        voff[i] = (unsigned)((row_of0(s) - wg_row0) * (C * 4) + q * 16);
    }
    const char* xb = reinterpret_cast<const char*>(x) + wg_row0 * (C * 4);
    char* ob = reinterpret_cast<char*>(out) + wg_row0 * (C * 4);
    const long step_bytes = (long)ps * (C * 4);
    auto step_pos = [&](int it) -> int { it = min(max(it, 0), nstep - 1); return dir ? (nstep - 1 - it) : it; };

    // weights of this pass: gate image and this wave's 16 output-projection columns, both resident in VGPRs
    f16x8 wh[4][4], wl[4][4];
    {
        const _Float16* wp = w_pk + ((long)(dir * 4 + wave) * 16 * 64 + lane) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wh[g][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(g * 4 + ks) * 64 * 16);
                wl[g][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(g * 4 + ks) * 64 * 16 + 8);
            }
    }
    f16x8 lwh[2], lwl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        lwh[ks] = *reinterpret_cast<const f16x8*>(&wlin_pk[((wave * 2 + ks) * 64 + lane) * 16]);
        lwl[ks] = *reinterpret_cast<const f16x8*>(&wlin_pk[((wave * 2 + ks) * 64 + lane) * 16 + 8]);
    }
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = b_sum[dir * 256 + g * 64 + unit];
    const float lbias = accumulate ? 0.0f : blin[unit];       // output bias rides in the accumulator (first pass only)

    // LDS addresses (in halves / floats) of this thread's roles; the double-buffer index is a compile-time constant of
    // the 2x unrolled step loop, so all of them are immediates on top of these bases
    int a_row[MT], l_row[MT];                       // row-wise role: row (tid >> 4), float4 q
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        a_row[i] = ((tid + 256 * i) >> 4) * LH_AP + q * 4;
        l_row[i] = ((tid + 256 * i) >> 4) * LSP + q * 4;
    }
    const int a_frag = l15 * LH_AP + g4 * 8;        // MFMA A fragment: row l15 (+16 m), halves g4*8 (+32 ks)
    const int a_cell = (g4 * 4) * LH_AP + C + unit; // cell role: rows g4*4 + r (+16 m), hidden column `unit`
    const int l_cell = (g4 * 4) * LSP + unit;

    auto load_x = [&](int it, float4 (&xr)[MT]) {
        const char* base = xb + step_pos(it) * step_bytes;
#pragma unroll
        for (int i = 0; i < MT; ++i) xr[i] = *reinterpret_cast<const float4*>(base + voff[i]);
    };
    auto store_split4 = [&](int idx, float a, float b, float c, float d) {
        f16x4 h4, l4;
        _Float16 th, tl;
        split_f16(a, th, tl); h4[0] = th; l4[0] = tl;
        split_f16(b, th, tl); h4[1] = th; l4[1] = tl;
        split_f16(c, th, tl); h4[2] = th; l4[2] = tl;
        split_f16(d, th, tl); h4[3] = th; l4[3] = tl;
        *reinterpret_cast<f16x4*>(&ahi[idx]) = h4;
        *reinterpret_cast<f16x4*>(&alo[idx]) = l4;
    };
    auto norm_store_x = [&](int buf, float4 (&xr)[MT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 v = xr[i];
            const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
            const float rstd = __builtin_amdgcn_rsqf(var + LN_EPS);      // var + eps >= 1e-5: no denormal guard needed
            store_split4(buf * NS * LH_AP + a_row[i], v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd);
        }
    };
    // base of the projection's accumulation for the rows of step `it`: pass 0 the residual (= the un-normalised
    // LSTM input itself), pass 1 the partial sum written by pass 0 (same thread, same rows)
    auto load_base = [&](int it, float4 (&rr)[MT]) {
        const char* base = (accumulate ? const_cast<const char*>(ob) : xb) + step_pos(it) * step_bytes;
#pragma unroll
        for (int i = 0; i < MT; ++i) rr[i] = *reinterpret_cast<const float4*>(base + voff[i]);
    };
    // finished rows of step `it`: base + (bias) + projection parked in ls[buf]
    auto store_rows = [&](int it, int buf, const float4 (&rr)[MT]) {
        char* base = ob + step_pos(it) * step_bytes;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const float4 pv = *reinterpret_cast<const float4*>(&ls[buf * NS * LSP + l_row[i]]);
            *reinterpret_cast<float4*>(base + voff[i]) =
                make_float4(rr[i].x + pv.x, rr[i].y + pv.y, rr[i].z + pv.z, rr[i].w + pv.w);
        }
    };
    // P = h W_lin^T for the h tile in A buffer `buf` -> ls[lbuf]   (k-steps 2,3 of the A rows are the h part)
    auto lin_tile = [&](int buf, int lbuf) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            f32x4 am = f32x4{lbias, lbias, lbias, lbias};
            const int ro = (buf * NS + m * 16) * LH_AP + a_frag;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[ro + (2 + ks) * 32]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[ro + (2 + ks) * 32]);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwh[ks], am, 0, 0, 0);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwl[ks], am, 0, 0, 0);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, lwh[ks], am, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) ls[(lbuf * NS + m * 16 + r) * LSP + l_cell] = am[r];
        }
    };

    // ---- prologue
    float creg[MT][4];                               // cell state
    float4 xr[MT], rr[MT];
    {
        load_x(0, xr);
        norm_store_x(0, xr);
        load_x(1, xr);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int s = min(s0 + ((tid + 256 * i) >> 4), nseq - 1);
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (h0) hv = *reinterpret_cast<const float4*>(&h0[(long)s * H + q * 4]);
            store_split4(a_row[i] + C, hv.x, hv.y, hv.z, hv.w);
            rr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = min(s0 + m * 16 + g4 * 4 + r, nseq - 1);
                creg[m][r] = c0 ? c0[(long)s * H + unit] : 0.0f;
            }
    }
    __syncthreads();

    // one step; CUR = A / projection buffer of this step (compile-time: the loop below is unrolled by two)
    auto step = [&](int it, auto cur_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        LH_STAMP(0);
        // The row-wise work of the step (x_{it+1} normalised and split into the free buffer, rows of step it-2 finished
        // and stored, the next rows fetched) is independent of the step's MFMAs, and a wave is strictly in-order: placed
        // in front of the MFMAs it costs ~640 cycles of a ~2900-cycle step (s_memtime trace, profiles/r02d_*), placed
        // BETWEEN them it issues in the matrix pipe's shadow (an MFMA keeps the pipe busy ~19 cycles and the issue port
        // ~12).  Four groups of 12 MFMAs (one k-step each), each with a slice of the row work; the fences between the
        // groups keep the memory operations in the order  consumers of last step's loads -> store -> new loads  so that
        // the wave never waits for a store acknowledge.
        f32x4 gate[MT][4];
        f16x8 fah[MT][4], fal[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gate[m][g] = f32x4{bias[g], bias[g], bias[g], bias[g]};
        }
        auto read_frag = [&](int ks) __attribute__((always_inline)) {     // fragments are fetched two k-steps ahead of their MFMAs
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int ro = (cur * NS + m * 16) * LH_AP + a_frag;
                fah[m][ks] = *reinterpret_cast<const f16x8*>(&ahi[ro + ks * 32]);
                fal[m][ks] = *reinterpret_cast<const f16x8*>(&alo[ro + ks * 32]);
            }
        };
        read_frag(0);
        read_frag(1);
        auto mfma_ks = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
#pragma unroll
                for (int g = 0; g < 4; ++g) gate[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[m][ks], wh[g][ks], gate[m][g], 0, 0, 0);
#if !defined(LH_PROBE_HIHI)
#pragma unroll
                for (int g = 0; g < 4; ++g) gate[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah[m][ks], wl[g][ks], gate[m][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < 4; ++g) gate[m][g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fal[m][ks], wh[g][ks], gate[m][g], 0, 0, 0);
#endif
            }
        };
        // group 0: k-step 0  +  x_{it+1}: statistics
#pragma unroll
        for (int i = 0; i < MT; ++i) pin_here(xr[i]);     // (keeps the unrolled twin's consumers behind its barrier)
        mfma_ks(0);
        read_frag(2);
        float4 xc[MT];
        float rstd[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 v = xr[i];
            const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
            rstd[i] = __builtin_amdgcn_rsqf(var + LN_EPS);      // var + eps >= 1e-5: no denormal guard needed
            xc[i] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
        // group 1: k-step 1  +  x_{it+1} split into the free buffer, rows of step it-2 summed up
        mfma_ks(1);
        read_frag(3);
        float4 done[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            store_split4(nxt * NS * LH_AP + a_row[i], xc[i].x * rstd[i], xc[i].y * rstd[i], xc[i].z * rstd[i], xc[i].w * rstd[i]);
            pin_here(rr[i]);
            const float4 pv = *reinterpret_cast<const float4*>(&ls[nxt * NS * LSP + l_row[i]]);
            done[i] = make_float4(rr[i].x + pv.x, rr[i].y + pv.y, rr[i].z + pv.z, rr[i].w + pv.w);
        }
        __builtin_amdgcn_sched_barrier(0);
        // group 2: k-step 2  +  this step's own global traffic
        mfma_ks(2);
#if defined(LH_PROBE_NOSTORE)
        if (it >= 2 && done[0].x == 1.2345e30f) {
#else
        if (it >= 2) {
#endif
            char* base = ob + step_pos(it - 2) * step_bytes;
#pragma unroll
            for (int i = 0; i < MT; ++i) *reinterpret_cast<float4*>(base + voff[i]) = done[i];
        }
#if defined(LH_PROBE_NOLOAD)
        if (it < 2) {
            load_base(it - 1, rr);
            load_x(it + 2, xr);
        }
#else
        load_base(it - 1, rr);                        // consumed next iteration (clamped at it = 0: unused)
        load_x(it + 2, xr);
#endif
        __builtin_amdgcn_sched_barrier(0);
        LH_STAMP(1);
        // group 3: k-step 3 and the projection
        mfma_ks(3);
        lin_tile(cur, cur);                           // projection of h_{it-1} (at it = 0: of the initial state, unused)
        __builtin_amdgcn_sched_barrier(0);
        LH_STAMP(2);
        __builtin_amdgcn_sched_barrier(0);

#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float hv;
                lstm_cell_pre(gate[m][0][r], gate[m][1][r], gate[m][2][r], gate[m][3][r], creg[m][r], hv);
                _Float16 th, tl;
                split_f16(hv, th, tl);
                ahi[(nxt * NS + m * 16 + r) * LH_AP + a_cell] = th;
                alo[(nxt * NS + m * 16 + r) * LH_AP + a_cell] = tl;
            }
        __builtin_amdgcn_sched_barrier(0);
        LH_STAMP(3);
        __syncthreads();
    };
    {
        int it = 0;
        for (; it + 1 < nstep; it += 2) {
            step(it, std::integral_constant<int, 0>{});
            step(it + 1, std::integral_constant<int, 1>{});
        }
        if (it < nstep) step(it, std::integral_constant<int, 0>{});
    }

#if defined(LH_PROBE_TRACE)
    if (tr_slot >= 0 && dir == 0 && threadIdx.x < 64)
        for (int i = threadIdx.x; i < 128 * 4; i += 64) lh_trace_buf[tr_slot * 512 + i] = tr[i];
#endif
    // ---- drain: rows of the last two steps
    if (nstep >= 2) store_rows(nstep - 2, (nstep - 1) & 1, rr);
    load_base(nstep - 1, rr);
    lin_tile(nstep & 1, nstep & 1);                   // projection of h_{nstep-1}
    __syncthreads();
    store_rows(nstep - 1, nstep & 1, rr);
}

#endif  // LH_LEGACY

// ------------------------------------------------------------------------------------------------------
// Eight-wave, software-pipelined fused recurrence for the inter pass (k_lstm_lin8p): 625 dependent steps and, at batch 32,
// only 194 sixteen-sequence tiles for 256 CUs, i.e. one workgroup per CU.  Same tile (16 sequences, all steps) and LDS
// images as k_ln_lstm_lin, but 512 threads, two waves per SIMD.  Wave v owns hidden units 8v..8v+7 of all four gates;
// the gate GEMM is TRANSPOSED (weights are the MFMA A operand, activations the B operand) so that the accumulator tile
// is [16 gate columns] x [16 sequences] with rows ordered (unit, gate): a lane then holds the four gates of ONE unit of
// ONE sequence in the four registers of a tile and the cell update stays lane-local with two cells per lane.  Waves
// 0..3 ("LIN") also run the output projection of h_{t-1} (their B fragments of the h half double as its A operand) and
// finish / fetch the output rows; waves 4..7 normalise and split x.  Each SIMD hosts one wave of either kind.
// The NON-recurrent half of the gate GEMM — LN(x_{t+1}) W_ih'^T + b, K = 64 of the 128 — is off the dependent chain:
// it is computed during step t, after the recurrent half of step t, so the chain of a step is
//     barrier -> ds_read h -> 12 MFMA (K = 64, accumulators start from the pre-computed x half) -> cells -> ds_write h.
// The x half of the LDS rows therefore runs one step further ahead than the h half: during step t buffer (t & 1)
// receives x_{t+2} (its x_t was consumed in step t-1) while x_{t+1} is read from buffer (t+1) & 1; the h halves
// alternate as usual.  Measured 1.70 ms per bench step (3 launches) against 1.82 for the un-pipelined form.
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512, 1) k_lstm_lin8p(const float* __restrict__ x, const _Float16* __restrict__ w_pk,
                                                       const float* __restrict__ b_sum, const _Float16* __restrict__ wlin_pk,
                                                       const float* __restrict__ blin, const float* __restrict__ h0,
                                                       const float* __restrict__ c0, float* __restrict__ hN,
                                                       float* __restrict__ cN, float* out, int nseq, int nstep, int sdiv,
                                                       int so, int si, int ps, int dir, int accumulate) {
    constexpr int NS = 16;
    constexpr int LSP = C + 4;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * NS * LH_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * NS * LH_AP];
    __shared__ __attribute__((aligned(16))) float ls[2 * NS * LSP];
    __shared__ __attribute__((aligned(16))) float hf[NS * LSP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s0 = blockIdx.x * NS;
    const int g4 = lane >> 4, l15 = lane & 15;
    const bool lin_wave = wave < 4;
    const int q = tid & 15, rrow = (tid & 255) >> 4;
    const int unit0 = 8 * wave + 2 * g4;             // adjacent units per lane (tile m = even / odd units): weights.py pack_lstm_f16x3_w8

    auto row_of0 = [&](int s) -> long { return (long)(s / sdiv) * so + (long)(s % sdiv) * si; };
    const long wg_row0 = row_of0(min(s0, nseq - 1));
    const unsigned voff = (unsigned)((row_of0(min(s0 + rrow, nseq - 1)) - wg_row0) * (C * 4) + q * 16);
    const char* xb = reinterpret_cast<const char*>(x) + wg_row0 * (C * 4);
    char* ob = reinterpret_cast<char*>(out) + wg_row0 * (C * 4);
    const char* bsrc = reinterpret_cast<const char*>(
        accumulate ? reinterpret_cast<unsigned long long>(ob) : reinterpret_cast<unsigned long long>(xb));
    const long step_bytes = (long)ps * (C * 4);
    auto step_pos = [&](int it) -> int { it = min(max(it, 0), nstep - 1); return dir ? (nstep - 1 - it) : it; };

    f16x8 wh[2][4], wl[2][4];
    {
        const _Float16* wp = w_pk + ((long)(dir * 8 + wave) * 8 * 64 + lane) * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wh[m][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(m * 4 + ks) * 64 * 16);
                wl[m][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(m * 4 + ks) * 64 * 16 + 8);
            }
    }
    f16x8 lwh[2], lwl[2];
    float lbias = 0.0f;
    if (lin_wave) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            lwh[ks] = *reinterpret_cast<const f16x8*>(&wlin_pk[((wave * 2 + ks) * 64 + lane) * 16]);
            lwl[ks] = *reinterpret_cast<const f16x8*>(&wlin_pk[((wave * 2 + ks) * 64 + lane) * 16 + 8]);
        }
        if (!accumulate) lbias = blin[wave * 16 + l15];
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lwh[ks] = lwl[ks] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float bias[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[m][g] = b_sum[dir * 256 + g * 64 + unit0 + m];

    const int a_frag = l15 * LH_AP + g4 * 8;
    const int a_cell = l15 * LH_AP + C + unit0;
    const int a_row = rrow * LH_AP + q * 4;
    const int l_row = rrow * LSP + q * 4;
    const int l_lin = (g4 * 4) * LSP + wave * 16 + l15;

    auto store_split4 = [&](int idx, float a, float b, float c, float d) {
        f16x4 h4, l4;
        _Float16 th, tl;
        split_f16(a, th, tl); h4[0] = th; l4[0] = tl;
        split_f16(b, th, tl); h4[1] = th; l4[1] = tl;
        split_f16(c, th, tl); h4[2] = th; l4[2] = tl;
        split_f16(d, th, tl); h4[3] = th; l4[3] = tl;
        *reinterpret_cast<f16x4*>(&ahi[idx]) = h4;
        *reinterpret_cast<f16x4*>(&alo[idx]) = l4;
    };
    auto norm_store_x = [&](int buf, float4 v) {
        const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
        const float rstd = __builtin_amdgcn_rsqf(var + LN_EPS);
        store_split4(buf * NS * LH_AP + a_row, v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd);
    };
    auto load_row = [&](const char* base, int it) -> float4 {
        return *reinterpret_cast<const float4*>(base + step_pos(it) * step_bytes + voff);
    };
    // x half of one step's gates: bias + LN(x) W_ih'^T from the x fragments (k-steps 0, 1) of A buffer `buf`
    auto x_half = [&](int buf, f32x4 (&g)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m) g[m] = f32x4{bias[m][0], bias[m][1], bias[m][2], bias[m][3]};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 xh = *reinterpret_cast<const f16x8*>(&ahi[buf * NS * LH_AP + a_frag + ks * 32]);
            const f16x8 xl = *reinterpret_cast<const f16x8*>(&alo[buf * NS * LH_AP + a_frag + ks * 32]);
#pragma unroll
            for (int m = 0; m < 2; ++m) g[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[m][ks], xh, g[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) g[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[m][ks], xl, g[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) g[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[m][ks], xh, g[m], 0, 0, 0);
        }
    };

    // ---- prologue: x_0, x_1 normalised into the two buffers, x_2 in flight; h_{-1}; x half of step 0
    float creg[2], hreg[2];
    float4 carry = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!lin_wave) {
        norm_store_x(0, load_row(xb, 0));
        norm_store_x(1, load_row(xb, 1));
        carry = load_row(xb, 2);
    } else {
        const int s = min(s0 + rrow, nseq - 1);
        float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h0) hv = *reinterpret_cast<const float4*>(&h0[(long)s * H + q * 4]);
        store_split4(a_row + C, hv.x, hv.y, hv.z, hv.w);
    }
    {
        const int s = min(s0 + l15, nseq - 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            creg[m] = c0 ? c0[(long)s * H + unit0 + m] : 0.0f;
            hreg[m] = 0.0f;
        }
    }
    __syncthreads();
    f32x4 gx[2];
    x_half(0, gx);
    __syncthreads();                                  // buffer 0's x half is rewritten (x_2) in step 0

    // One step, specialised on the wave's role (LIN: waves 0..3) and on the buffer parity, as ONE basic block behind the
    // row-wise prologue so that the scheduler can be told to interleave the off-chain MFMAs with the cell update.
    auto step = [&](int it, auto cur_tag, auto lin_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        constexpr bool LIN = decltype(lin_tag)::value;
        // recurrent half first: acc = (x half of this step, computed a step ago) + h_{it-1} W_hh^T ...
        f16x8 bh[2], bl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bh[ks] = *reinterpret_cast<const f16x8*>(&ahi[cur * NS * LH_AP + a_frag + (2 + ks) * 32]);
            bl[ks] = *reinterpret_cast<const f16x8*>(&alo[cur * NS * LH_AP + a_frag + (2 + ks) * 32]);
        }
        f32x4 acc[2] = {gx[0], gx[1]};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[m][2 + ks], bh[ks], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[m][2 + ks], bl[ks], acc[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[m][2 + ks], bh[ks], acc[m], 0, 0, 0);
        }
        // ... with the wave's row-wise role in the shadow of those MFMAs (in-order issue: in front of them it would delay
        // the chain by its full length): consumers of the global loads issued one step ago, then this step's own traffic
        pin_here(carry);
        if (LIN) {
            const float4 pv = *reinterpret_cast<const float4*>(&ls[nxt * NS * LSP + l_row]);
            const float4 done = make_float4(carry.x + pv.x, carry.y + pv.y, carry.z + pv.z, carry.w + pv.w);
            if (it >= 2) *reinterpret_cast<float4*>(ob + step_pos(it - 2) * step_bytes + voff) = done;
            carry = load_row(bsrc, it - 1);
        } else {
            norm_store_x(cur, carry);                 // x_{it+2}
            carry = load_row(xb, it + 3);
        }
        __builtin_amdgcn_sched_barrier(0);
        // Off the chain: x half of step it+1 (buffer nxt holds x_{it+1}) and, on waves 0..3, the projection of h_{it-1}.
        // The two waves of a SIMD take opposite orders: the LIN wave issues its 18 off-chain MFMAs first and updates its
        // cells afterwards, the other wave updates its cells first.  (A SIMD issues matrix and vector instructions from
        // one port and they do not overlap — profiles/r02d_lstm_issue_probes.txt; cutting the cell update into
        // micro-stages behind each off-chain MFMA measured 1.84 ms per step against 1.70 for this form.)
        auto cells = [&]() {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                lstm_cell_pre(acc[m][0], acc[m][1], acc[m][2], acc[m][3], creg[m], hreg[m]);
                _Float16 th, tl;
                split_f16(hreg[m], th, tl);
                ahi[nxt * NS * LH_AP + a_cell + m] = th;
                alo[nxt * NS * LH_AP + a_cell + m] = tl;
            }
        };
        if (LIN) {
            x_half(nxt, gx);
            f32x4 am = f32x4{lbias, lbias, lbias, lbias};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], lwh[ks], am, 0, 0, 0);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[ks], lwl[ks], am, 0, 0, 0);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[ks], lwh[ks], am, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            cells();
#pragma unroll
            for (int r = 0; r < 4; ++r) ls[cur * NS * LSP + r * LSP + l_lin] = am[r];
        } else {
            __builtin_amdgcn_sched_barrier(0);
            cells();
            __builtin_amdgcn_sched_barrier(0);
            x_half(nxt, gx);
        }
        __builtin_amdgcn_sched_barrier(0);            // keep the off-chain MFMAs on this side of the barrier
        __syncthreads();
    };
    auto run = [&](auto lin_tag) __attribute__((always_inline)) {
        int it = 0;
        for (; it + 1 < nstep; it += 2) {
            step(it, std::integral_constant<int, 0>{}, lin_tag);
            step(it + 1, std::integral_constant<int, 1>{}, lin_tag);
        }
        if (it < nstep) step(it, std::integral_constant<int, 0>{}, lin_tag);
    };
    if (lin_wave) run(std::true_type{});
    else run(std::false_type{});

    // ---- drain: rows of the last two steps (projection of h_{nstep-1} still to do)
    const int lastb = nstep & 1;
    if (lin_wave) {
        if (nstep >= 2) {
            const float4 pv = *reinterpret_cast<const float4*>(&ls[(lastb ^ 1) * NS * LSP + l_row]);
            *reinterpret_cast<float4*>(ob + step_pos(nstep - 2) * step_bytes + voff) =
                make_float4(carry.x + pv.x, carry.y + pv.y, carry.z + pv.z, carry.w + pv.w);
        }
        carry = load_row(bsrc, nstep - 1);
        f32x4 am = f32x4{lbias, lbias, lbias, lbias};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[lastb * NS * LH_AP + a_frag + (2 + ks) * 32]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[lastb * NS * LH_AP + a_frag + (2 + ks) * 32]);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwh[ks], am, 0, 0, 0);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwl[ks], am, 0, 0, 0);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, lwh[ks], am, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ls[lastb * NS * LSP + r * LSP + l_lin] = am[r];
    }
    if (hN) {
#pragma unroll
        for (int m = 0; m < 2; ++m) hf[l15 * LSP + unit0 + m] = hreg[m];
    }
    __syncthreads();
    if (lin_wave) {
        const float4 pv = *reinterpret_cast<const float4*>(&ls[lastb * NS * LSP + l_row]);
        *reinterpret_cast<float4*>(ob + step_pos(nstep - 1) * step_bytes + voff) =
            make_float4(carry.x + pv.x, carry.y + pv.y, carry.z + pv.z, carry.w + pv.w);
    } else if (hN && s0 + rrow < nseq) {
        *reinterpret_cast<float4*>(&hN[(long)(s0 + rrow) * H + q * 4]) = *reinterpret_cast<const float4*>(&hf[l_row]);
    }
    if (cN && s0 + l15 < nseq) {
#pragma unroll
        for (int m = 0; m < 2; ++m) cN[(long)(s0 + l15) * H + unit0 + m] = creg[m];
    }
}

#if defined(LH_LEGACY)
static int g_dephase = 1;          // lh_set_tuning(3, 0) switches the slot-parity issue priority off (A/B runs)
template <int MT>
static int launch_lstm_lin(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                           const float* h0, const float* c0, float* hN, float* cN, float* out, int nseq, int nstep,
                           int sdiv, int so, int si, int ps, int dir, int accumulate, hipStream_t st) {
    constexpr int NS = 16 * MT;
    // k_ln_lstm_lin serves the intra pass (zero initial state, no state out): it never writes hN / cN, so a caller that
    // asks for them must be refused instead of handed uninitialised memory
    if (h0 || c0 || hN || cN) return LH_ERR_ARG;
    hipLaunchKernelGGL((k_ln_lstm_lin<MT>), dim3((nseq + NS - 1) / NS), dim3(256), 0, st, x, (const _Float16*)w_pk, b_sum,
                       (const _Float16*)wlin_pk, blin, h0, c0, hN, cN, out, nseq, nstep, sdiv, so, si, ps, dir, accumulate,
                       g_dephase);
    return check_launch();
}

#endif  // LH_LEGACY
template <int MT>
static int launch_lstm_h3(const float* x, const float* lnw, const float* lnb, const void* w_pk, const float* b_sum,
                          const float* h0, const float* c0, float* hN, float* cN, float* h_out, int nseq, int nstep,
                          int ndir, int sdiv, int so, int si, int ps, int ldh, hipStream_t st) {
    constexpr int NS = 16 * MT;
    hipLaunchKernelGGL((k_ln_lstm_h3<MT>), dim3((nseq + NS - 1) / NS, ndir), dim3(256), 0, st, x, lnw, lnb,
                       (const _Float16*)w_pk, b_sum, h0, c0, hN, cN, h_out, nseq, nstep, sdiv, so, si, ps, ldh);
    return check_launch();
}

template <int MT>
static int launch_lstm(const float* x, const float* lnw, const float* lnb, const float* w_pk, const float* b_sum,
                       const float* h0, const float* c0, float* hN, float* cN, float* h_out, int nseq, int nstep,
                       int ndir, int sdiv, int so, int si, int ps, int ldh, hipStream_t st) {
    constexpr int NS = 16 * MT;
    hipLaunchKernelGGL((k_ln_lstm<MT>), dim3((nseq + NS - 1) / NS, ndir), dim3(256), 0, st, x, lnw, lnb, w_pk, b_sum,
                       h0, c0, hN, cN, h_out, nseq, nstep, sdiv, so, si, ps, ldh);
    return check_launch();
}

}  // namespace lh

namespace lh {
static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}
int emb_set(int key, int value);        // lh_embed.hip
static int g_tune[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // [0] intra MT (0 = auto), [1] inter MT (0 = auto), [3] 0 = no issue-priority de-phasing
}
namespace lh { int attn_set_mq(int v); }      // lh_attn.hip
namespace lh {                                // lh_recur.hip
int launch_intra_xp(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin, float* out,
                    int nseq, int nstep, int sdiv, int so, int si, int ps, int dir, int accumulate, hipStream_t st);
int xp_set(int key, int v);
int launch_inter_xp(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                    const float* h0, const float* c0, float* hN, float* cN, float* out, int nseq, int nstep, int sdiv, int so,
                    int si, int ps, hipStream_t st, int cflags = 0);
}
namespace lh { int backend_set_runs(int v); } // lh_backend.hip
#if defined(LH_PROBE_TRACE)
extern "C" int lh_probe_trace_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::lh_trace_buf), sizeof(lh::lh_trace_buf)) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int lh_set_tuning(int key, int value) {
    if (key == 4) return lh::attn_set_mq(value);
    if (key == 6) return lh::backend_set_runs(value);
    if (key >= 7 && key < 16) return lh::xp_set(key, value);      // lh_recur.hip switches
    if (key == 16 || key == 17) return lh::emb_set(key, value);   // lh_embed.hip: k_emb_rec issue priority, attention GEMM variant
    if (key < 0 || key >= 8) return LH_ERR_ARG;
#if defined(LH_LEGACY)
    if (key == 3) lh::g_dephase = value;
#else
    if (key == 2 && value == 2) return LH_ERR_UNSUPPORTED;       // k_ln_lstm_lin: lab builds only (-DLH_LEGACY)
#endif
    lh::g_tune[key] = value;
    return LH_OK;
}

extern "C" int lh_ln_lstm_intra(const float* x, const float* ln_w, const float* ln_b, const void* w_pk,
                                const float* b_sum, float* h_out, int n_frames, int mode, lh_stream_t stream) {
    using namespace lh;
    if (!x || !ln_w || !ln_b || !w_pk || !b_sum || !h_out || n_frames <= 0) return LH_ERR_ARG;
    // sequence s = frame (b,t); step p = frequency bin; row(s,p) = s*97 + p
    const int mt = g_tune[0] ? g_tune[0] : (mode == LH_GEMM_F16X3 ? 0 : (n_frames >= 8192 ? 2 : 1));
    if (mode == LH_GEMM_F16X3) {
        hipStream_t st = (hipStream_t)stream;
        if (mt == 2)
            return launch_lstm_h3<2>(x, ln_w, ln_b, w_pk, b_sum, nullptr, nullptr, nullptr, nullptr, h_out, n_frames, NF,
                                     2, 1, NF, 0, 1, 2 * H, st);
        if (mt == 1)
            return launch_lstm_h3<1>(x, ln_w, ln_b, w_pk, b_sum, nullptr, nullptr, nullptr, nullptr, h_out, n_frames, NF,
                                     2, 1, NF, 0, 1, 2 * H, st);
        // automatic: 32-sequence workgroups are ~18 % cheaper per sequence (matrix / VALU overlap across the two
        // tiles of a wave) but a launch is only as fast as its last round of workgroups, so whole rounds (2 resident
        // workgroups per CU) run as 32-sequence tiles and the remainder as 16-sequence tiles in a second launch.
        const int slots = 2 * cu_count();
        const int tiles2 = (n_frames / 32) * 2 / slots * slots / 2;      // per direction; x2 directions = whole rounds
        const int f2 = tiles2 * 32;
        int rc = LH_OK;
        if (f2 > 0)
            rc = launch_lstm_h3<2>(x, ln_w, ln_b, w_pk, b_sum, nullptr, nullptr, nullptr, nullptr, h_out, f2, NF, 2, 1, NF,
                                   0, 1, 2 * H, st);
        if (rc == LH_OK && n_frames > f2)
            rc = launch_lstm_h3<1>(x + (long)f2 * NF * C, ln_w, ln_b, w_pk, b_sum, nullptr, nullptr, nullptr, nullptr,
                                   h_out + (long)f2 * NF * 2 * H, n_frames - f2, NF, 2, 1, NF, 0, 1, 2 * H, st);
        return rc;
    }
    if (mode != LH_GEMM_F32) return LH_ERR_UNSUPPORTED;
    const float* w_f32 = (const float*)w_pk;
    if (mt == 2)
        return launch_lstm<2>(x, ln_w, ln_b, w_f32, b_sum, nullptr, nullptr, nullptr, nullptr, h_out, n_frames, NF, 2, 1,
                              NF, 0, 1, 2 * H, (hipStream_t)stream);
    return launch_lstm<1>(x, ln_w, ln_b, w_f32, b_sum, nullptr, nullptr, nullptr, nullptr, h_out, n_frames, NF, 2, 1, NF,
                          0, 1, 2 * H, (hipStream_t)stream);
}

// time window of the unfused intra pass (batches below the fused kernels' size): frames (b, t0 + j), one 16-sequence tile
// shape (a window of a small batch is a single round of workgroups anyway)
extern "C" int lh_ln_lstm_intra_win(const float* x, const float* ln_w, const float* ln_b, const void* w_pk, const float* b_sum,
                                    float* h_out, int B, int T, int t0, int Tc, lh_stream_t stream) {
    using namespace lh;
    if (!x || !ln_w || !ln_b || !w_pk || !b_sum || !h_out || B <= 0 || T <= 0 || t0 < 0 || Tc <= 0 || t0 + Tc > T) return LH_ERR_ARG;
    const long off = (long)t0 * NF;
    return launch_lstm_h3<1>(x + off * C, ln_w, ln_b, w_pk, b_sum, nullptr, nullptr, nullptr, nullptr, h_out + off * 2 * H, B * Tc,
                             NF, 2, Tc, T * NF, NF, 1, 2 * H, (hipStream_t)stream);
}

extern "C" int lh_ln_lstm_inter(const float* x, const float* ln_w, const float* ln_b, const void* w_pk,
                                const float* b_sum, const float* h0, const float* c0, float* hN, float* cN,
                                float* h_out, int B, int T, int mode, lh_stream_t stream) {
    using namespace lh;
    if (!x || !ln_w || !ln_b || !w_pk || !b_sum || !h0 || !c0 || !hN || !cN || !h_out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (h0 == hN || c0 == cN) return LH_ERR_ARG;
    // sequence s = b*97 + f; step p = frame t; row(s,p) = (b*T + p)*97 + f
    const int nseq = B * NF;
    const int mt = g_tune[1] == 2 ? 2 : 1;      // 32-sequence tiles only on request (lh_set_tuning(1, 2): A/B runs, tests/test_emu_kernels.py)
    if (mode == LH_GEMM_F16X3) {
        if (mt == 2)
            return launch_lstm_h3<2>(x, ln_w, ln_b, w_pk, b_sum, h0, c0, hN, cN, h_out, nseq, T, 1, NF, T * NF, 1, NF, H,
                                     (hipStream_t)stream);
        return launch_lstm_h3<1>(x, ln_w, ln_b, w_pk, b_sum, h0, c0, hN, cN, h_out, nseq, T, 1, NF, T * NF, 1, NF, H,
                                 (hipStream_t)stream);
    }
    if (mode != LH_GEMM_F32) return LH_ERR_UNSUPPORTED;
    const float* w_f32 = (const float*)w_pk;
    if (mt == 2)
        return launch_lstm<2>(x, ln_w, ln_b, w_f32, b_sum, h0, c0, hN, cN, h_out, nseq, T, 1, NF, T * NF, 1, NF, H,
                              (hipStream_t)stream);
    return launch_lstm<1>(x, ln_w, ln_b, w_f32, b_sum, h0, c0, hN, cN, h_out, nseq, T, 1, NF, T * NF, 1, NF, H,
                          (hipStream_t)stream);
}

// Fused variants: LayerNorm + (Bi)LSTM + output Linear + residual (split-precision mode only).
extern "C" int lh_intra_block(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk,
                              const float* blin, float* out, int n_frames, lh_stream_t stream) {
    using namespace lh;
    if (!x || !w_pk || !b_sum || !wlin_pk || !blin || !out || n_frames <= 0 || x == out) return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    // sequence = frame (b,t), step = frequency bin; forward launch then reverse launch (accumulating)
    int rc = LH_OK;
    for (int dir = 0; dir < 2 && rc == LH_OK; ++dir)
#if defined(LH_LEGACY)
        if (g_tune[2] == 2)                       // the previous fused kernel (A/B lab builds only)
            rc = launch_lstm_lin<1>(x, w_pk, b_sum, (const _Float16*)wlin_pk + (long)dir * 4 * 2 * 64 * 16, blin, nullptr, nullptr,
                                    nullptr, nullptr, out, n_frames, NF, 1, NF, 0, 1, dir, dir, st);
        else
#endif
            // software-pipelined, hand-ordered step (lh_recur.hip)
            rc = launch_intra_xp(x, w_pk, b_sum, (const _Float16*)wlin_pk + (long)dir * 4 * 2 * 64 * 16, blin, out, n_frames,
                                 NF, 1, NF, 0, 1, dir, dir, st);
    return rc;
}

// Time windows (ABI 14): the same fused stages on frames [t0, t0 + Tc) of every utterance of [B][T][97][64] buffers — the
// causal structure of the block (tfgridnet_causal.py:505-538: the intra pass is per frame, the inter pass carries (h, c))
// lets a host cut the time axis and run block i on window k+1 beside block i+1 on window k (net.py `time_chunks`).
extern "C" int lh_intra_block_win(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk,
                                  const float* blin, float* out, int B, int T, int t0, int Tc, lh_stream_t stream) {
    using namespace lh;
    if (!x || !w_pk || !b_sum || !wlin_pk || !blin || !out || B <= 0 || T <= 0 || x == out || t0 < 0 || Tc <= 0 || t0 + Tc > T)
        return LH_ERR_ARG;
    // sequence s = (b, j): row base (b * T + t0 + j) * 97, step = frequency bin
    const long off = (long)t0 * NF * C;
    int rc = LH_OK;
    for (int dir = 0; dir < 2 && rc == LH_OK; ++dir)
        rc = launch_intra_xp(x + off, w_pk, b_sum, (const _Float16*)wlin_pk + (long)dir * 4 * 2 * 64 * 16, blin, out + off,
                             B * Tc, NF, Tc, T * NF, NF, 1, dir, dir, (hipStream_t)stream);
    return rc;
}

extern "C" int lh_inter_block_win(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk,
                                  const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out,
                                  int B, int T, int t0, int Tc, int carry, lh_stream_t stream) {
    using namespace lh;
    if (!x || !w_pk || !b_sum || !wlin_pk || !blin || !h0 || !c0 || !hN || !cN || !out || B <= 0 || T <= 0) return LH_ERR_ARG;
    if (h0 == hN || c0 == cN || x == out || t0 < 0 || Tc < 2 || t0 + Tc > T || (carry & ~3)) return LH_ERR_ARG;
    // sequence s = b*97 + f; step j = frame t0 + j; row(s, j) = (b*T + t0 + j)*97 + f
    const long off = (long)t0 * NF * C;
    return launch_inter_xp(x + off, w_pk, b_sum, wlin_pk, blin, h0, c0, hN, cN, out + off, B * NF, Tc, NF, T * NF, 1, NF,
                           (hipStream_t)stream, carry);
}

extern "C" int lh_inter_block(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk,
                              const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out,
                              int B, int T, lh_stream_t stream) {
    using namespace lh;
    if (!x || !w_pk || !b_sum || !wlin_pk || !blin || !h0 || !c0 || !hN || !cN || !out || B <= 0 || T <= 0) return LH_ERR_ARG;
    if (h0 == hN || c0 == cN || x == out) return LH_ERR_ARG;
    // sequence s = b*97 + f; step = frame t; row(s, t) = (b*T + t)*97 + f.  Eight-wave tiles (k_lstm_lin8p): the pass is a
    // 625-step dependent chain, two waves per SIMD cover each other's latencies
    const int nseq = B * NF;
    // T >= 2: the hand-ordered step with per-phase issue priority (k_inter_xp, lh_recur.hip: its software pipeline needs a
    // second step); a single frame (streaming chunks) runs k_lstm_lin8p below; lh_set_tuning(5, 2) forces it (A/B)
    if (g_tune[5] != 2 && T >= 2)
        return launch_inter_xp(x, w_pk, b_sum, wlin_pk, blin, h0, c0, hN, cN, out, nseq, T, NF, T * NF, 1, NF,
                               (hipStream_t)stream);
    hipLaunchKernelGGL(k_lstm_lin8p, dim3((nseq + 15) / 16), dim3(512), 0, (hipStream_t)stream, x, (const _Float16*)w_pk,
                           b_sum, (const _Float16*)wlin_pk, blin, h0, c0, hN, cN, out, nseq, T, NF, T * NF, 1, NF, 0, 0);
    return check_launch();
}
