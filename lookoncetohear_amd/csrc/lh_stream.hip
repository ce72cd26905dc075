// Low-latency intra-frame BiLSTM for the streaming / small-batch case (BASELINE configs[1]: batch-1, 8 ms chunks).
//
// With one frame per utterance in flight there is ONE real sequence per (utterance, direction): the tiled MFMA
// recurrence (lh_lstm.hip) then runs a 16-row tile with 15 dead rows and pays ~1.1 us per step for 54 matrix
// instructions and a workgroup barrier — 97 steps x 3 blocks = 0.32 of the 0.60 ms chunk.  Here a workgroup owns one
// (frame, direction) and treats the step as what it is, a 256 x 64 mat-vec:
//   1. the input half of all 97 steps is hoisted out of the recurrence: G_x = LN(x) W_ih'^T + b as ONE
//      [97 x 64] x [64 x 256] split-precision MFMA GEMM into LDS (LayerNorm affine folded into W_ih', b);
//   2. 512 threads = 256 gate rows x 2 halves of the k range: a thread keeps its half row of W_hh (32 fp32 registers)
//      and adds it to h_{t-1} with plain fp32 FMAs, h read as LDS broadcasts; the rows are laid out so that the two
//      halves and the four gates of a hidden unit sit in one 16-lane row — DPP row rotations add the halves and bring
//      the gates together, 8 lanes per wave update the cell, ONE barrier per step.
// Reference: tfgridnet_causal.py:505-512 (intra_norm + intra_rnn); output in the unfused layout [rows][128] consumed
// by lh_linear_res.
#include "lh_split.h"

namespace lh {

constexpr int IS_GP = 4 * H;               // 256 gate columns
constexpr int IS_NT = 512;                 // threads: gate column x half of the k range
constexpr int IS_NLD = (NF * 16 + IS_NT - 1) / IS_NT;

// grid = n_frames * 2 (direction = blockIdx & 1), block 512
__global__ void __launch_bounds__(IS_NT) k_intra_stream(const float* __restrict__ x, const _Float16* __restrict__ wih_pk,
                                                        const float* __restrict__ b_sum, const float* __restrict__ whh,
                                                        float* __restrict__ h_out, int n_frames) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[FR_A];          // LN(x) of the frame, A image [97 -> 112 x 64]
    __shared__ __attribute__((aligned(16))) _Float16 alo[FR_A];
    __shared__ __attribute__((aligned(16))) float gxs[NF * IS_GP];       // input half of the gates, [step][column]
    __shared__ __attribute__((aligned(16))) float hs[2][H];              // h_{t-1} / h_t
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int frame = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* xf = x + (long)frame * NF * C;

    // Thread tid = 64 w + 16 r + 8 u + 2 g + kh owns gate g of hidden unit 8 w + 2 r + u for the k range
    // [32 kh, 32 kh + 32): its half row of W_hh stays in 32 registers (weights.py lays the rows out in thread order).
    const int kh = tid & 1;
    float wr[H / 2];
    {
        const float* wp = whh + ((long)dir * IS_NT + tid) * (H / 2);
#pragma unroll
        for (int k4 = 0; k4 < H / 8; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(wp + k4 * 4);
            wr[k4 * 4 + 0] = v.x; wr[k4 * 4 + 1] = v.y; wr[k4 * 4 + 2] = v.z; wr[k4 * 4 + 3] = v.w;
        }
    }

    // ---- LayerNorm over the 64 channels of every (frame, f) row (affine folded into the weights), split to fp16 hi/lo
    for (int i = tid; i < FR_A; i += IS_NT) { ahi[i] = (_Float16)0.f; alo[i] = (_Float16)0.f; }
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

    // ---- G_x[p][n] = b[n] + LN(x)[p] . W_ih'[n]: wave w owns column tiles 2w, 2w+1 (columns in (tid >> 1) order)
#pragma unroll 1
    for (int i = 0; i < 2; ++i) {
        const int nt = 2 * wave + i;
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
    float c = 0.f;
    const int unit = wave * 8 + g4 * 2 + (l15 >> 3);
    float* hrow = h_out + ((long)frame * NF) * 2 * H + dir * H + unit;
    for (int it = 0; it < NF; ++it) {
        const int p = dir ? NF - 1 - it : it;
        const float* hp = hs[it & 1] + kh * (H / 2);
        float a0 = kh ? 0.f : gxs[p * IS_GP + (tid >> 1)], a1 = 0.f, a2 = 0.f, a3 = 0.f;
        // this half of h_{t-1} into registers first (two addresses per wave: broadcast reads), ONE wait, then the FMAs:
        // left to itself the scheduler interleaves read / wait / 4 FMAs and exposes the LDS latency every time
        float4 hreg[H / 8];
#pragma unroll
        for (int k4 = 0; k4 < H / 8; ++k4) hreg[k4] = *reinterpret_cast<const float4*>(hp + k4 * 4);
#if defined(__AMDGCN__)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int k4 = 0; k4 < H / 8; ++k4) {
            a0 = fmaf(wr[k4 * 4 + 0], hreg[k4].x, a0);
            a1 = fmaf(wr[k4 * 4 + 1], hreg[k4].y, a1);
            a2 = fmaf(wr[k4 * 4 + 2], hreg[k4].z, a2);
            a3 = fmaf(wr[k4 * 4 + 3], hreg[k4].w, a3);
        }
        const float part = (a0 + a1) + (a2 + a3);
        // DPP row rotations (a few cycles each, no LDS crossbar): the other k half sits one lane up, the gates f, g, o of
        // this lane's unit 2, 4, 6 lanes up
        const float gate = part + row_ror_mov<15>(part);
        const float gf = row_ror_mov<14>(gate), gg = row_ror_mov<12>(gate), go = row_ror_mov<10>(gate);
        if ((l15 & 7) == 0) {
            float hv;
            lstm_cell(gate, gf, gg, go, c, hv);
            hs[(it + 1) & 1][unit] = hv;
            hrow[(long)p * 2 * H] = hv;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------------
// Inter-frame LSTM for few sequences (batch 1-5 offline): LayerNorm -> causal LSTM over time with carried (h, c) ->
// Linear(64->64) -> + residual  (tfgridnet_causal.py:521-538), one workgroup per sequence (b, f).
// The tiled kernel (k_ln_lstm_lin) needs 16 sequences per workgroup: at batch 1 that is 7 workgroups walking 625
// steps at ~1.1 us each (0.7 ms per block, 2.1 of the 3.0 ms forward).  Here every sequence gets its own CU and the
// step is the same 256 x 64 mat-vec as in k_intra_stream; the time axis is cut into chunks of 64 steps whose input
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
                                                        float* __restrict__ out, int T) {
    __shared__ __attribute__((aligned(16))) _Float16 xhi[IM_A];          // LN(x) of the chunk's steps
    __shared__ __attribute__((aligned(16))) _Float16 xlo[IM_A];
    __shared__ __attribute__((aligned(16))) _Float16 hhi[IM_A];          // h_t of the chunk's steps
    __shared__ __attribute__((aligned(16))) _Float16 hlo[IM_A];
    __shared__ __attribute__((aligned(16))) float xraw[IM_TC * IM_XP];   // un-normalised rows (residual)
    __shared__ __attribute__((aligned(16))) float gxs[IM_TC * IS_GP];    // input half of the gates, [step][column]
    __shared__ __attribute__((aligned(16))) float hs[2][H];              // h_{t-1} / h_t
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int seq = blockIdx.x, b = seq / NF, f = seq % NF;              // state row b*97 + f
    const float* xs = x + ((long)b * T * NF + f) * C;                    // step t -> + t * 97 * 64
    float* os = out + ((long)b * T * NF + f) * C;
    const long tstride = (long)NF * C;

    const int kh = tid & 1;
    float wr[H / 2];
    {
        const float* wp = whh + (long)tid * (H / 2);
#pragma unroll
        for (int k4 = 0; k4 < H / 8; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(wp + k4 * 4);
            wr[k4 * 4 + 0] = v.x; wr[k4 * 4 + 1] = v.y; wr[k4 * 4 + 2] = v.z; wr[k4 * 4 + 3] = v.w;
        }
    }
    const int unit = wave * 8 + g4 * 2 + (l15 >> 3);
    const bool cell_lane = (l15 & 7) == 0;
    float c = cell_lane ? c0[(long)seq * H + unit] : 0.f;
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
        for (int i = 0; i < 2; ++i) {
            const int nt = 2 * wave + i;
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
        for (int it = 0; it < nst; ++it) {
            const float* hp = hs[hb] + kh * (H / 2);
            float a0 = kh ? 0.f : gxs[it * IS_GP + (tid >> 1)], a1 = 0.f, a2 = 0.f, a3 = 0.f;
            float4 hreg[H / 8];
#pragma unroll
            for (int k4 = 0; k4 < H / 8; ++k4) hreg[k4] = *reinterpret_cast<const float4*>(hp + k4 * 4);
#if defined(__AMDGCN__)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int k4 = 0; k4 < H / 8; ++k4) {
                a0 = fmaf(wr[k4 * 4 + 0], hreg[k4].x, a0);
                a1 = fmaf(wr[k4 * 4 + 1], hreg[k4].y, a1);
                a2 = fmaf(wr[k4 * 4 + 2], hreg[k4].z, a2);
                a3 = fmaf(wr[k4 * 4 + 3], hreg[k4].w, a3);
            }
            const float part = (a0 + a1) + (a2 + a3);
            const float gate = part + row_ror_mov<15>(part);
            const float gf = row_ror_mov<14>(gate), gg = row_ror_mov<12>(gate), go = row_ror_mov<10>(gate);
            if (cell_lane) {
                float hv;
                lstm_cell(gate, gf, gg, go, c, hv);
                hs[hb ^ 1][unit] = hv;
                _Float16 hh, hl_;
                split_hl(hv, hh, hl_);
                const int idx = a_index<IM_TC>(it, unit);
                hhi[idx] = hh;
                hlo[idx] = hl_;
            }
            hb ^= 1;
            __syncthreads();
        }
        // ---- projection + residual: out[step][o] = x[step][o] + b[o] + sum_u h[step][u] W_lin[o][u]; 16 (row tile,
        //      column tile) products over the 8 waves
#pragma unroll 1
        for (int p = wave; p < (IM_TC / 16) * 4; p += 8) {
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
    if (cell_lane) cN[(long)seq * H + unit] = c;
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

extern "C" int lh_inter_matvec(const float* x, const void* wih_pk, const float* b_sum, const float* whh, const void* wlin_pk,
                               const float* blin, const float* h0, const float* c0, float* hN, float* cN, float* out,
                               int B, int T, lh_stream_t stream) {
    using namespace lh;
    if (!x || !wih_pk || !b_sum || !whh || !wlin_pk || !blin || !h0 || !c0 || !hN || !cN || !out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (h0 == hN || c0 == cN || x == out) return LH_ERR_ARG;
    hipLaunchKernelGGL(k_inter_matvec, dim3(B * NF), dim3(IS_NT), 0, (hipStream_t)stream, x, (const _Float16*)wih_pk, b_sum,
                       whh, (const _Float16*)wlin_pk, blin, h0, c0, hN, cN, out, T);
    return check_launch();
}
