// Low-latency intra-frame BiLSTM for the streaming / small-batch case (BASELINE configs[1]: batch-1, 8 ms chunks).
//
// With one frame per utterance in flight there is ONE real sequence per (utterance, direction): the tiled MFMA
// recurrence (lh_lstm.hip) then runs a 16-row tile with 15 dead rows and pays ~1.1 us per step for 54 matrix
// instructions and a workgroup barrier — 97 steps x 3 blocks = 0.32 of the 0.60 ms chunk.  Here a workgroup owns one
// (frame, direction) and treats the step as what it is, a 256 x 64 mat-vec:
//   1. the input half of all 97 steps is hoisted out of the recurrence: G_x = LN(x) W_ih'^T + b as ONE
//      [97 x 64] x [64 x 256] split-precision MFMA GEMM into LDS (LayerNorm affine folded into W_ih', b);
//   2. thread n keeps row n of W_hh (64 fp32 registers) and adds W_hh[n] . h_{t-1} with plain fp32 FMAs, h broadcast
//      from LDS; the rows are permuted so that the four gates of a hidden unit sit 4 lanes apart inside one 16-lane
//      row — three DPP row rotations bring them together, 16 lanes per wave update the cell, ONE barrier per step.
// Reference: tfgridnet_causal.py:505-512 (intra_norm + intra_rnn); output in the unfused layout [rows][128] consumed
// by lh_linear_res.
#include "lh_split.h"

namespace lh {

constexpr int IS_GP = 4 * H;               // 256 gate columns

// grid = n_frames * 2 (direction = blockIdx & 1), block 256
__global__ void __launch_bounds__(256) k_intra_stream(const float* __restrict__ x, const _Float16* __restrict__ wih_pk,
                                                      const float* __restrict__ b_sum, const float* __restrict__ whh,
                                                      float* __restrict__ h_out, int n_frames) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[FR_A];          // LN(x) of the frame, A image [97 -> 112 x 64]
    __shared__ __attribute__((aligned(16))) _Float16 alo[FR_A];
    __shared__ __attribute__((aligned(16))) float gxs[NF * IS_GP];       // input half of the gates, [step][thread]
    __shared__ __attribute__((aligned(16))) float hs[2][H];              // h_{t-1} / h_t
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    const int frame = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const float* xf = x + (long)frame * NF * C;

    // this thread's row of W_hh (gate (lane & 15) >> 2 of hidden unit 16 wave + 4 (lane >> 4) + (lane & 3); weights.py
    // permutes the rows into thread order)
    float wr[H];
    {
        const float* wp = whh + ((long)dir * IS_GP + tid) * H;
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(wp + k4 * 4);
            wr[k4 * 4 + 0] = v.x; wr[k4 * 4 + 1] = v.y; wr[k4 * 4 + 2] = v.z; wr[k4 * 4 + 3] = v.w;
        }
    }

    // ---- LayerNorm over the 64 channels of every (frame, f) row (affine folded into the weights), split to fp16 hi/lo
    frame_zero_pad(ahi, alo, tid);
    {
        float4 stg[FR_NLD];
        frame_load(xf, tid, stg);
#pragma unroll
        for (int i = 0; i < FR_NLD; ++i) {
            const int e = tid + 256 * i;           // 16 threads per row; rows beyond 96 are clamped copies (not stored)
            float4 v = stg[i];
            const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
            v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
            const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
            const float rstd = rsqrtf(var + LN_EPS);
            if (e < NF * 16)
                store_split4<FR_RP>(ahi, alo, e >> 4, (e & 15) * 4, make_float4(v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd));
        }
    }
    if (tid < H) hs[0][tid] = 0.f;
    __syncthreads();

    // ---- G_x[p][n] = b[n] + LN(x)[p] . W_ih'[n]: wave w owns column tiles 4w .. 4w+3 (its own threads' columns)
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        const int nt = 4 * wave + i;
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
    const int unit = wave * 16 + g4 * 4 + (l15 & 3);          // this lane's hidden unit; its gate is l15 >> 2
    float* hrow = h_out + ((long)frame * NF) * 2 * H + dir * H + unit;
    for (int it = 0; it < NF; ++it) {
        const int p = dir ? NF - 1 - it : it;
        const float* hp = hs[it & 1];
        float a0 = gxs[p * IS_GP + tid], a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < H / 4; ++k4) {
            const float4 h4 = *reinterpret_cast<const float4*>(hp + k4 * 4);     // same address in every lane: broadcast
            a0 = fmaf(wr[k4 * 4 + 0], h4.x, a0);
            a1 = fmaf(wr[k4 * 4 + 1], h4.y, a1);
            a2 = fmaf(wr[k4 * 4 + 2], h4.z, a2);
            a3 = fmaf(wr[k4 * 4 + 3], h4.w, a3);
        }
        const float gate = (a0 + a1) + (a2 + a3);
        // gates f, g, o of this lane's unit sit 4, 8, 12 lanes up inside the 16-lane row: three DPP row rotations
        // (a few cycles each) instead of a trip through the LDS crossbar
        const float gf = row_ror_mov<12>(gate), gg = row_ror_mov<8>(gate), go = row_ror_mov<4>(gate);
        if (l15 < 4) {
            float hv;
            lstm_cell(gate, gf, gg, go, c, hv);
            hs[(it + 1) & 1][unit] = hv;
            hrow[(long)p * 2 * H] = hv;
        }
        __syncthreads();
    }
}

}  // namespace lh

extern "C" int lh_intra_stream(const float* x, const void* wih_pk, const float* b_sum, const float* whh, float* h_out,
                               int n_frames, lh_stream_t stream) {
    using namespace lh;
    if (!x || !wih_pk || !b_sum || !whh || !h_out || n_frames <= 0) return LH_ERR_ARG;
    hipLaunchKernelGGL(k_intra_stream, dim3(2 * n_frames), dim3(256), 0, (hipStream_t)stream, x, (const _Float16*)wih_pk,
                       b_sum, whh, h_out, n_frames);
    return check_launch();
}
