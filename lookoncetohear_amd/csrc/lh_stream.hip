// Low-latency intra-frame BiLSTM for the streaming / small-batch case (BASELINE configs[1]: batch-1, 8 ms chunks).
//
// With one frame per utterance in flight there is ONE real sequence per (utterance, direction): the tiled MFMA
// recurrence (lh_lstm.hip) then runs a 16-row tile with 15 dead rows and pays ~1.1 us per step for 54 matrix
// instructions and a workgroup barrier — 97 steps x 3 blocks = 0.32 of the 0.60 ms chunk.  Here a workgroup owns one
// (frame, direction) and treats the step as what it is, a 256 x 64 mat-vec:
//   1. the input half of all 97 steps is hoisted out of the recurrence: G_x = LN(x) W_ih'^T + b as ONE
//      [97 x 64] x [64 x 256] split-precision MFMA GEMM into LDS (LayerNorm affine folded into W_ih', b);
//   2. the step is latency, not throughput: ONE wave per SIMD (256 threads) and four lanes per hidden unit.  Lane
//      (unit u, slice s) keeps the 16-wide k slice s of the unit's four gate rows of W_hh (64 fp32 registers), reads its
//      16 values of h_{t-1} from LDS (4 broadcast reads instead of the 8 of a half row) and does the 64 FMAs as 32 packed
//      ones (v_pk_fma_f32 on whole register pairs: the safe form, build.py).  The four partial sums per gate are then
//      REDUCE-SCATTERED over the quad with DPP quad permutes (the gate slots of a lane are XOR-rotated by its slice
//      index, so lane s ends up with the full sum of gate s after three DPP adds, against 12 adds for an all-reduce),
//      so every lane evaluates ONE gate non-linearity (2 transcendentals per lane and step instead of 8 on one lane in
//      eight), and two more quad permutes bring i*g and o to the lane that holds the cell state.  4 transcendental + ~70 other instructions per step on an otherwise empty SIMD, one barrier:
//      0.49 -> 0.36-0.38 us per step (profiles/r03k_*).
// Reference: tfgridnet_causal.py:505-512 (intra_norm + intra_rnn); output in the unfused layout [rows][128] consumed
// by lh_linear_res.
#include "lh_quad.h"

namespace lh {

constexpr int IS_NLD = (NF * 16 + IS_NT - 1) / IS_NT;

// grid = n_frames * 2 (direction = blockIdx & 1), block 512
__global__ void __launch_bounds__(IS_NT) k_intra_stream(const float* __restrict__ x, const _Float16* __restrict__ wih_pk,
                                                        const float* __restrict__ b_sum, const float* __restrict__ whh,
                                                        float* __restrict__ h_out, int n_frames) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[FR_A];          // LN(x) of the frame, A image [97 -> 112 x 64]
    __shared__ __attribute__((aligned(16))) _Float16 alo[FR_A];
    __shared__ __attribute__((aligned(16))) float gxs[NF * IS_GP];       // input half of the gates, [step][column]
    __shared__ __attribute__((aligned(16))) float hs[2][H];              // h_{t-1} / h_t
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int frame = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* xf = x + (long)frame * NF * C;

    // Thread tid = 4 u + s < 256: hidden unit u, k slice s (quad_step)
    const int unit = (tid & (IS_NR - 1)) >> 2, qs = tid & 3;
    f32x2 wr[4][8];
    quad_load_w(whh + (long)dir * IS_GP * H, unit, qs, wr);

    // ---- LayerNorm over the 64 channels of every (frame, f) row (affine folded into the weights), split to fp16 hi/lo
    static_assert(FR_A % 8 == 0, "16-byte zero fill");
    for (int i = tid; i < FR_A / 8; i += IS_NT) {
        *reinterpret_cast<f16x8*>(&ahi[i * 8]) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        *reinterpret_cast<f16x8*>(&alo[i * 8]) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < IS_NLD; ++i) {
        const int e = tid + IS_NT * i, ec = min(e, NF * 16 - 1);       // 16 threads per row; clamped copies are not stored
        float4 v = *reinterpret_cast<const float4*>(&xf[(ec >> 4) * C + (ec & 15) * 4]);
        const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
        const float rstd = rsqrtf(var + LN_EPS);
        if (e < NF * 16)
            store_split4<FR_RP>(ahi, alo, e >> 4, (e & 15) * 4, make_float4(v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd));
    }
    if (tid < H) hs[0][tid] = 0.f;
    __syncthreads();

    // ---- G_x[p][n] = b[n] + LN(x)[p] . W_ih'[n]: wave w owns column tiles 2w, 2w+1
#pragma unroll 1
    for (int i = 0; i < 16 / IS_NW; ++i) {
        const int nt = (16 / IS_NW) * wave + i;
        f16x8 wh[2], wl[2];
        load_w<2>(wih_pk + (long)dir * 16 * 2 * 64 * 16, nt, lane, wh, wl);
        const float bz = b_sum[dir * IS_GP + nt * 16 + l15];
#pragma unroll 1
        for (int m = 0; m < FR_RP / 16; ++m) {
            const f32x4 acc = mma_tile<FR_RP, 2>(ahi, alo, m, g4, l15, wh, wl, bz);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = m * 16 + g4 * 4 + r;
                if (p < NF) gxs[p * IS_GP + nt * 16 + l15] = acc[r];
            }
        }
    }
    __syncthreads();

    // ---- recurrence: zero initial state (the intra LSTM carries nothing between frames)
    if (tid >= IS_NR) {
        for (int it = 0; it < NF; ++it) QS_SYNC();
        return;
    }
    float c = 0.f;                                      // (scaled by QS_K2; zero either way)
    const float gscale = quad_gate_scale(qs);
    float* hrow = h_out + ((long)frame * NF) * 2 * H + dir * H + unit;
    for (int it = 0; it < NF; ++it) {
        const int p = dir ? NF - 1 - it : it;
        const float gx = gscale * gxs[p * IS_GP + tid];
        const float hv = quad_step(wr, hs[it & 1] + 16 * qs, gx, c, qs);
        if (qs == 1) {
            hs[(it + 1) & 1][unit] = hv;
            hrow[(long)p * 2 * H] = hv;
        }
        QS_SYNC();
    }
}

// ------------------------------------------------------------------------------------------------------
// Inter-frame LSTM for few sequences (batch 1-2 offline since round 6; the kernel serves any number): LayerNorm -> causal LSTM over time with carried (h, c) ->
// Linear(64->64) -> + residual  (tfgridnet_causal.py:521-538), one workgroup per sequence (b, f).
// The tiled kernel (k_ln_lstm_lin) needs 16 sequences per workgroup: at batch 1 that is 7 workgroups walking 625
// steps at ~1.1 us each (0.7 ms per block, 2.1 of the 3.0 ms forward).  Here every sequence gets its own CU and the
// step is the same quad-lane 256 x 64 mat-vec as in k_intra_stream (quad_step); the time axis is cut into chunks of 64 steps whose input
// half (LN(x) W_ih'^T + b) and output projection (h W_lin^T + b + residual) run as split-precision MFMA GEMMs before
// and after the chunk's recurrence.
// ------------------------------------------------------------------------------------------------------
constexpr int IM_TC = 64;                  // steps per chunk
constexpr int IM_A = 2 * 4 * IM_TC * 8;    // halves per [64 x 64] A image
constexpr int IM_XP = C + 4;               // raw-x staging row

__global__ void __launch_bounds__(IS_NT) k_inter_matvec(const float* __restrict__ x, const _Float16* __restrict__ wih_pk,
                                                        const float* __restrict__ b_sum, const float* __restrict__ whh,
                                                        const _Float16* __restrict__ wlin_pk, const float* __restrict__ blin,
                                                        const float* __restrict__ h0, const float* __restrict__ c0,
                                                        float* __restrict__ hN, float* __restrict__ cN,
                                                        float* __restrict__ out, int T, int Tfull, int tw0, int cflags) {
    // time window (lh_inter_matvec_win): steps 0 .. T-1 of the launch are frames tw0 .. tw0+T-1 of buffers laid out for Tfull
    // frames; cflags bit 0 = c0 holds the kernel's scaled cell state (QS_K2 c, written by the previous window with bit 1),
    // bit 1 = cN is written that way (like k_inter_xp: the windows then reproduce the whole-clip launch bit for bit when
    // they start on multiples of the 64-step chunk)
    __shared__ __attribute__((aligned(16))) _Float16 xhi[IM_A];          // LN(x) of the chunk's steps
    __shared__ __attribute__((aligned(16))) _Float16 xlo[IM_A];
    __shared__ __attribute__((aligned(16))) _Float16 hhi[IM_A];          // h_t of the chunk's steps
    __shared__ __attribute__((aligned(16))) _Float16 hlo[IM_A];
    __shared__ __attribute__((aligned(16))) float xraw[IM_TC * IM_XP];   // un-normalised rows (residual)
    __shared__ __attribute__((aligned(16))) float gxs[IM_TC * IS_GP];    // input half of the gates, [step][column]
    __shared__ __attribute__((aligned(16))) float hs[2][H];              // h_{t-1} / h_t
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int seq = blockIdx.x, b = seq / NF, f = seq % NF;              // state row b*97 + f
    const float* xs = x + (((long)b * Tfull + tw0) * NF + f) * C;        // step t -> + t * 97 * 64
    float* os = out + (((long)b * Tfull + tw0) * NF + f) * C;
    const long tstride = (long)NF * C;

    const int unit = (tid & (IS_NR - 1)) >> 2, qs = tid & 3;            // hidden unit, k slice (quad_step): threads < 256
    f32x2 wr[4][8];
    quad_load_w(whh, unit, qs, wr);
    const bool cell_lane = qs == 1 && tid < IS_NR;
    float c = cell_lane ? ((cflags & 1) ? 1.0f : QS_K2) * c0[(long)seq * H + unit] : 0.f;       // carried times QS_K2
    const float gscale = quad_gate_scale(qs);
    if (tid < H) hs[0][tid] = h0[(long)seq * H + tid];
    int hb = 0;                                                          // hs buffer holding h_{t-1}

    for (int t0 = 0; t0 < T; t0 += IM_TC) {
        const int nst = min(IM_TC, T - t0);
        // ---- stage the chunk: raw rows (residual) and LayerNorm'ed rows as the x A image; 16 threads per row
#pragma unroll
        for (int i = 0; i < IM_TC * 16 / IS_NT; ++i) {
            const int e = tid + IS_NT * i, r = e >> 4, q4 = (e & 15) * 4;
            float4 v = *reinterpret_cast<const float4*>(&xs[(long)min(t0 + r, T - 1) * tstride + q4]);
            *reinterpret_cast<float4*>(&xraw[r * IM_XP + q4]) = v;
            const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
            const float rstd = rsqrtf(var + LN_EPS);
            store_split4<IM_TC>(xhi, xlo, r, q4, make_float4(v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd));
        }
        __syncthreads();
        // ---- G_x[step][n] = b[n] + LN(x)[step] . W_ih'[n]: wave w owns column tiles 2w, 2w+1
#pragma unroll 1
        for (int i = 0; i < 16 / IS_NW; ++i) {
            const int nt = (16 / IS_NW) * wave + i;
            f16x8 wh[2], wl[2];
            load_w<2>(wih_pk, nt, lane, wh, wl);
            const float bz = b_sum[nt * 16 + l15];
#pragma unroll 1
            for (int m = 0; m < IM_TC / 16; ++m) {
                const f32x4 acc = mma_tile<IM_TC, 2>(xhi, xlo, m, g4, l15, wh, wl, bz);
#pragma unroll
                for (int r = 0; r < 4; ++r) gxs[(m * 16 + g4 * 4 + r) * IS_GP + nt * 16 + l15] = acc[r];
            }
        }
        __syncthreads();
        // ---- recurrence over the chunk's steps (see k_intra_stream); h_t also goes into the projection's A image
        if (tid >= IS_NR) {
            for (int it = 0; it < nst; ++it) QS_SYNC();
            hb ^= nst & 1;
        } else {
            for (int it = 0; it < nst; ++it) {
                const float gx = gscale * gxs[it * IS_GP + tid];
                const float hv = quad_step(wr, hs[hb] + 16 * qs, gx, c, qs);
                if (cell_lane) {
                    hs[hb ^ 1][unit] = hv;
                    _Float16 hh, hl_;
                    split_hl(hv, hh, hl_);
                    const int idx = a_index<IM_TC>(it, unit);
                    hhi[idx] = hh;
                    hlo[idx] = hl_;
                }
                hb ^= 1;
                QS_SYNC();
            }
        }
        // ---- projection + residual: out[step][o] = x[step][o] + b[o] + sum_u h[step][u] W_lin[o][u]; 16 (row tile,
        //      column tile) products over the waves
#pragma unroll 1
        for (int p = wave; p < (IM_TC / 16) * 4; p += IS_NW) {
            const int mt = p >> 2, nt = p & 3;
            f16x8 wh[2], wl[2];
            load_w<2>(wlin_pk, nt, lane, wh, wl);
            const f32x4 acc = mma_tile<IM_TC, 2>(hhi, hlo, mt, g4, l15, wh, wl, blin[nt * 16 + l15]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int st = mt * 16 + g4 * 4 + r;
                if (st < nst) os[(long)(t0 + st) * tstride + nt * 16 + l15] = acc[r] + xraw[st * IM_XP + nt * 16 + l15];
            }
        }
        __syncthreads();
    }
    if (tid < H) hN[(long)seq * H + tid] = hs[hb][tid];
    if (cell_lane) cN[(long)seq * H + unit] = c * ((cflags & 2) ? 1.0f : 1.0f / QS_K2);
}

}  // namespace lh

extern "C" int lh_intra_stream(const float* x, const void* wih_pk, const float* b_sum, const float* whh, float* h_out,
                               int n_frames, lh_stream_t stream) {
    using namespace lh;
    if (!x || !wih_pk || !b_sum || !whh || !h_out || n_frames <= 0) return LH_ERR_ARG;
    hipLaunchKernelGGL(k_intra_stream, dim3(2 * n_frames), dim3(IS_NT), 0, (hipStream_t)stream, x, (const _Float16*)wih_pk,
                       b_sum, whh, h_out, n_frames);
    return check_launch();
}

extern "C" int lh_inter_matvec_win(const float* x, const void* wih_pk, const float* b_sum, const float* whh, const void* wlin_pk,
                                   const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out,
                                   int B, int T, int t0, int Tc, int carry, lh_stream_t stream) {
    using namespace lh;
    if (!x || !wih_pk || !b_sum || !whh || !wlin_pk || !blin || !h0 || !c0 || !hN || !cN || !out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (h0 == hN || c0 == cN || x == out || t0 < 0 || Tc <= 0 || t0 + Tc > T || (carry & ~3)) return LH_ERR_ARG;
    hipLaunchKernelGGL(k_inter_matvec, dim3(B * NF), dim3(IS_NT), 0, (hipStream_t)stream, x, (const _Float16*)wih_pk, b_sum,
                       whh, (const _Float16*)wlin_pk, blin, h0, c0, hN, cN, out, Tc, T, t0, carry);
    return check_launch();
}

extern "C" int lh_inter_matvec(const float* x, const void* wih_pk, const float* b_sum, const float* whh, const void* wlin_pk,
                               const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out,
                               int B, int T, lh_stream_t stream) {
    return lh_inter_matvec_win(x, wih_pk, b_sum, whh, wlin_pk, blin, h0, c0, hN, cN, out, B, T, 0, T, 0, stream);
}
