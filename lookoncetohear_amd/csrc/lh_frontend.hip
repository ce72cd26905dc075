// Front end of the separator: STFT analysis + causal 3x3 conv, and the speaker-gain projection.
// HBM-streaming kernels (<2 % of the path's FLOPs): one workgroup owns a tile of
// frames, keeps samples and the spectrum tile in LDS and the filterbank in registers.
#include "lh_split.h"

namespace lh {

// ---- STFT analysis + causal 3x3 conv on split-precision fp16 MFMA, persistent workgroups of 8 waves ---------------
// A tile = 14 output frames of one utterance (+2 halo frames of the causal conv = 16 STFT frames = one MFMA row tile per
// microphone).  Both contractions run as f16x3 products (v = hi + lo, lh_split.h): the exact-fp32 MFMA version of this
// kernel spent 38 k matrix cycles per tile and wave (conv 28 k, STFT 10 k) at one wave per SIMD — 0.23 ms per call at
// B = 32 against an HBM floor of 0.065 ms; this one needs 7.8 k.
//   * STFT: the 16 frames of a microphone are the A rows (K = 192 samples, fp16 hi/lo images in LDS, 200-half row
//     stride: conflict-free ds_read_b128); the analysis filterbank lives in VGPRs as B fragments, wave w owns filter-row
//     tiles w and w + 8 of 13 (96 registers).
//   * The spectrum tile is written BIN-MAJOR, specT[f + 1][frame * 4 + ch] (rows 0 and 98 = the zero padding of the
//     f -/+ 1 taps): the 12 taps (kt, ch) of output frame jt at frequency offset kf are then 12 CONTIGUOUS halves of row
//     f + kf starting at jt * 4, so the conv's A fragments are plain 8-byte-aligned LDS reads — no im2col.  The K axis
//     is 3 chunks of 16 (12 taps + 4 zero-weight slots that read the next frame's finite values) padded to 64.
//   * conv: per output frame [97 bins x 64] x [64 x 64 channels]; wave w owns channel tile w & 3 and the row tiles of
//     its parity; output rows leave through a double-buffered LDS stage as 256-byte coalesced stores, one barrier
//     per frame.
constexpr int FE_NTH = 512;
constexpr int FE_TT = 14;                      // output frames per tile
constexpr int FE_NJ = 16;                      // STFT frames per tile (2 halo + 14)
constexpr int FE_AP = NFFT + 8;                // 200 halves per frame row of the A images
constexpr int FE_SP = 68;                      // specT row: 16 frames x 4 ch + 4 pad halves (136 B: conflict-free ds_read_b64)
constexpr int FE_NT = (NK + 15) / 16;          // 13 filter-row tiles (194 -> 208)
constexpr int FE_KS = NFFT / 32;               // 6 STFT k-steps
constexpr int FE_OP = C + 4;                   // output staging row

__global__ void __launch_bounds__(FE_NTH, 1) k_stft_conv_in(const float* __restrict__ x, const float* __restrict__ cbuf_in,
                                                            float* __restrict__ cbuf_out, const _Float16* __restrict__ wfb_pk,
                                                            const _Float16* __restrict__ wc_pk, const float* __restrict__ bc,
                                                            float* __restrict__ z, int B, int T, int n_samples) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[NMIC * FE_NJ * FE_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[NMIC * FE_NJ * FE_AP];
    __shared__ __attribute__((aligned(16))) _Float16 sth[(NF + 2) * FE_SP];
    __shared__ __attribute__((aligned(16))) _Float16 stl[(NF + 2) * FE_SP];
    __shared__ __attribute__((aligned(16))) float outs[2][NF * FE_OP];
    // Range-safe splits (pow2_scale, lh_common.h): ONE power of two per tile, from the largest sample magnitude of its 16
    // frames (and 1/32 of the largest carried spectrum value: a filter row's L1 norm is < 23, so a spectrum value is below
    // 32 x the sample maximum), brings the samples into [2^9, 2^10) and the spectra below 2^15; both contractions are
    // linear, so the scale is undone once, on the conv accumulator (the bias joins after it).  A quiet recording (x 1e-4)
    // keeps its 22 bits, a hot one (x 1e3) does not overflow the fp16 halves.
    __shared__ float tmax[FE_NTH / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;

    // resident B fragments: filterbank tiles wave, wave + 8; conv weights of channel tile wave & 3
    f16x8 wfh[2][FE_KS], wfl[2][FE_KS];
#pragma unroll
    for (int i = 0; i < 2; ++i) load_w<FE_KS>(wfb_pk, min(wave + 8 * i, FE_NT - 1), lane, wfh[i], wfl[i]);
    f16x8 wch[2], wcl[2];
    const int cnt = wave & 3;
    load_w<2>(wc_pk, cnt, lane, wch, wcl);
    const float cbias = bc[cnt * 16 + l15];

    static_assert(((NF + 2) * FE_SP) % 4 == 0, "8-byte zero fill");
    for (int i = tid; i < (NF + 2) * FE_SP / 4; i += FE_NTH) {          // pad rows / columns
        *reinterpret_cast<f16x4*>(&sth[i * 4]) = f16x4{0, 0, 0, 0};
        *reinterpret_cast<f16x4*>(&stl[i * 4]) = f16x4{0, 0, 0, 0};
    }

    const int tiles_per_b = (T + FE_TT - 1) / FE_TT;
    for (int tile = blockIdx.x; tile < B * tiles_per_b; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile % tiles_per_b) * FE_TT;
        const int nt_out = min(FE_TT, T - t0);
        __syncthreads();                       // previous tile fully consumed (A images / specT / outs / tmax)
        float tsc, tinv;                       // this tile's scale and its inverse

        // stage the 16 frames of both microphones: frame j = samples (t0-2+j)*128 .. +192, 48 float4 each
        {
            constexpr int NLD = NMIC * FE_NJ * 48 / FE_NTH;    // 3
            float4 stg[NLD];
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + FE_NTH * i;
                const int m = e / (FE_NJ * 48), j = (e / 48) % FE_NJ, c4 = e % 48;
                const int t = t0 - 2 + j;
                stg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && t < T)
                    stg[i] = *reinterpret_cast<const float4*>(&x[((long)b * NMIC + m) * n_samples + (long)t * HOP + c4 * 4]);
            }
            float tm = 0.f;
#pragma unroll
            for (int i = 0; i < NLD; ++i) tm = fmaxf(tm, absmax4(stg[i]));
            if (t0 == 0)                               // carried halo frames enter the same spectrum tile (workgroup-uniform)
                for (int i = tid; i < 4 * 2 * NF; i += FE_NTH) tm = fmaxf(tm, fabsf(cbuf_in[(long)b * 4 * 2 * NF + i]) * (1.0f / 32.0f));
            tm = wave_max(tm);
            if (lane == 0) tmax[wave] = tm;
            __syncthreads();
#pragma unroll
            for (int w = 0; w < FE_NTH / 64; ++w) tm = fmaxf(tm, tmax[w]);
            pow2_scale<9>(tm, tsc, tinv);
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = tid + FE_NTH * i;
                const int m = e / (FE_NJ * 48), j = (e / 48) % FE_NJ, c4 = e % 48;
                const float v[4] = {stg[i].x * tsc, stg[i].y * tsc, stg[i].z * tsc, stg[i].w * tsc};
                f16x2_t h01, l01, h23, l23;
                split_pair(v[0], v[1], h01, l01);
                split_pair(v[2], v[3], h23, l23);
                const f16x4 h4 = f16x4{h01[0], h01[1], h23[0], h23[1]}, l4 = f16x4{l01[0], l01[1], l23[0], l23[1]};
                const int idx = (m * FE_NJ + j) * FE_AP + c4 * 4;
                *reinterpret_cast<f16x4*>(&ahi[idx]) = h4;
                *reinterpret_cast<f16x4*>(&alo[idx]) = l4;
            }
        }
        __syncthreads();

        // spectrum: [16 frames x 192] x [192 x 16 filter rows] per (mic, filter tile) -> specT (split), carried halo
#pragma unroll 1
        for (int m = 0; m < NMIC; ++m) {
            f32x4 am[2], ac[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { am[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            const bool second = wave + 8 < FE_NT;      // wave-uniform: waves 0..4 own two filter tiles
#pragma unroll
            for (int ks = 0; ks < FE_KS; ++ks) {
                const int idx = (m * FE_NJ + l15) * FE_AP + ks * 32 + g4 * 8;
                const f16x8 fah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
                const f16x8 fal = *reinterpret_cast<const f16x8*>(&alo[idx]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (i == 0 || second) {
                        am[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah, wfh[i][ks], am[i], 0, 0, 0);
                        ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah, wfl[i][ks], ac[i], 0, 0, 0);
                        ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fal, wfh[i][ks], ac[i], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int nt = wave + 8 * i;
                const int k = nt * 16 + l15;                           // filter row: k < 97 re, 97..193 im
                if (nt < FE_NT && k < NK) {
                    const int ch = (k / NF) * NMIC + m, f = k % NF;    // channels re_m0, re_m1, im_m0, im_m1
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = g4 * 4 + r, t = t0 - 2 + j;
                        float v = am[i][r] + ac[i][r];             // spectrum value times the tile scale
                        if (t < 0) v = cbuf_in[(((long)b * 4 + ch) * 2 + (t + 2)) * NF + f] * tsc;   // carried halo frames
                        if (t >= T) v = 0.0f;
                        _Float16 h, l;
                        split_hl(v, h, l);
                        sth[(f + 1) * FE_SP + j * 4 + ch] = h;
                        stl[(f + 1) * FE_SP + j * 4 + ch] = l;
                        // new halo state = the last two frames of the halo-extended spectrum (exact fp32, unscaled)
                        if (t >= T - 2 && t < T) cbuf_out[(((long)b * 4 + ch) * 2 + (t - (T - 2))) * NF + f] = v * tinv;
                    }
                }
            }
        }
        __syncthreads();

        // conv: per output frame [97 bins x 64 (3 chunks of 12 taps + zero-weight slots)] x [64 x 16 channels of this wave]
        const int mt0 = wave >> 2;                     // row tiles mt0, mt0 + 2, mt0 + 4 (, 6)
        const int kf0 = g4 >> 1, cofs = (g4 & 1) * 8;  // k-step 0: chunk kf0; k-step 1: chunk 2 (lanes g4 >= 2 hit zero weights)
        for (int jt = 0; jt < nt_out; ++jt) {
            float* ob = outs[jt & 1];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mt = mt0 + 2 * i;
                if (mt < 7) {                  // wave-uniform
                    const int f = min(mt * 16 + l15, NF - 1);          // rows >= 97 are dropped below
                    const int a0 = (f + kf0) * FE_SP + jt * 4 + cofs, a1 = (f + 2) * FE_SP + jt * 4 + cofs;
                    f16x8 ah[2], al[2];
                    auto rd8 = [&](const _Float16* base, int idx) -> f16x8 {     // 8 halves, 8-byte aligned
                        const f16x4 u = *reinterpret_cast<const f16x4*>(&base[idx]);
                        const f16x4 w = *reinterpret_cast<const f16x4*>(&base[idx + 4]);
                        return f16x8{u[0], u[1], u[2], u[3], w[0], w[1], w[2], w[3]};
                    };
                    ah[0] = rd8(sth, a0); al[0] = rd8(stl, a0);
                    ah[1] = rd8(sth, a1); al[1] = rd8(stl, a1);
                    f32x4 am = f32x4{0.f, 0.f, 0.f, 0.f}, ac = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], wch[ks], am, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], wcl[ks], ac, 0, 0, 0);
                        ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], wch[ks], ac, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int fo = mt * 16 + g4 * 4 + r;
                        if (fo < NF) ob[fo * FE_OP + cnt * 16 + l15] = fmaf(am[r] + ac[r], tinv, cbias);
                    }
                }
            }
            __syncthreads();                   // frame complete in outs[jt & 1]; the stage of frame jt - 1 is free again
            float* dst = z + (((long)b * T + t0 + jt) * NF) * C;
            for (int e = tid; e < NF * 16; e += FE_NTH)
                *reinterpret_cast<float4*>(&dst[e * 4]) = *reinterpret_cast<const float4*>(&ob[(e >> 4) * FE_OP + (e & 15) * 4]);
        }
    }
}

// speaker-gain projection, row-parallel: grid (ceil(6208/32), ceil(B/8)); each workgroup streams 32 rows of W once
// and applies them to 8 utterances (W is read from HBM once per launch, re-used from L2 across the batch groups)
constexpr int EP_ROWS = 32, EP_NB = 8;
__global__ void __launch_bounds__(256) k_embed_proj(const float* __restrict__ emb, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ raw, int B) {
    constexpr int N = C * NF;
    __shared__ __attribute__((aligned(16))) float es[EP_NB][SPK];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    const int b0 = blockIdx.y * EP_NB;
    for (int i = tid; i < EP_NB * SPK; i += 256) {
        const int bb = min(b0 + i / SPK, B - 1);
        es[i / SPK][i % SPK] = emb[(long)bb * SPK + (i % SPK)];
    }
    __syncthreads();
    // a wave's 8 rows of W and their biases are fetched up front (8 independent 16-byte loads per lane in flight; one row
    // per loop trip exposed the load latency 8 times; 17.8 -> 16.8 us per call, profiles/r05o_*)
    constexpr int RW = EP_ROWS / 4;
    float4 w4[RW];
    float bz[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const int r = min((int)blockIdx.x * EP_ROWS + wave + q * 4, N - 1);
        w4[q] = *reinterpret_cast<const float4*>(&w[(long)r * SPK + lane * 4]);
        bz[q] = bias[r];
    }
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const int r = blockIdx.x * EP_ROWS + wave + q * 4;
        if (r >= N) break;
#pragma unroll
        for (int j = 0; j < EP_NB; ++j) {
            const float4 e4 = *reinterpret_cast<const float4*>(&es[j][lane * 4]);
            float s = wave_sum(w4[q].x * e4.x + w4[q].y * e4.y + w4[q].z * e4.z + w4[q].w * e4.w);
            if (lane == 0 && b0 + j < B) raw[(long)(b0 + j) * N + r] = s + bz[q];
        }
    }
}

// LayerNorm over the 6208 projected values of one utterance + (c,f) -> (f,c) transpose; grid B
__global__ void __launch_bounds__(256) k_embed_ln(const float* __restrict__ raw, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, float* __restrict__ gain) {
    constexpr int N = C * NF;   // 6208
    __shared__ float vals[N];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    float s = 0.0f;
    for (int i = tid; i < N; i += 256) { const float v = raw[(long)b * N + i]; vals[i] = v; s += v; }
    const float mean = block_sum_256(s, red) * (1.0f / N);
    float v = 0.0f;
    for (int i = tid; i < N; i += 256) { float d = vals[i] - mean; v += d * d; }
    const float rstd = rsqrtf(block_sum_256(v, red) * (1.0f / N) + LN_EPS);
    // (the coalesced form — affine in i order back into LDS, transpose on the way out — measured 19.4-20.0 us against 17.4 for
    // this strided gather of lnw / lnb: one more barrier and LDS pass than the gather costs; profiles/r05o_*)
    for (int oidx = tid; oidx < N; oidx += 256) {
        const int f = oidx / C, c = oidx % C;
        const int i = c * NF + f;                     // reference flat order is channel-major (reshape [B,C,F])
        gain[(long)b * N + oidx] = (vals[i] - mean) * rstd * lnw[i] + lnb[i];
    }
}

}  // namespace lh

extern "C" int lh_stft_conv_in(const float* x, const float* conv_buf_in, float* conv_buf_out, const void* wfb_pk,
                               const void* wconv_pk, const float* bconv, float* z, int B, int T, int n_samples,
                               lh_stream_t stream) {
    using namespace lh;
    if (!x || !conv_buf_in || !conv_buf_out || !wfb_pk || !wconv_pk || !bconv || !z || B <= 0 || T <= 0) return LH_ERR_ARG;
    if (conv_buf_in == conv_buf_out || n_samples != T * HOP + (NFFT - HOP)) return LH_ERR_ARG;
    const int tiles = B * ((T + FE_TT - 1) / FE_TT);
    hipLaunchKernelGGL(k_stft_conv_in, dim3(tiles < 256 ? tiles : 256), dim3(FE_NTH), 0, (hipStream_t)stream, x,
                       conv_buf_in, conv_buf_out, (const _Float16*)wfb_pk, (const _Float16*)wconv_pk, bconv, z, B, T,
                       n_samples);
    return check_launch();
}

extern "C" int lh_embed_proj_ln(const float* emb, const float* w, const float* bias, const float* ln_w,
                                const float* ln_b, float* scratch, float* gain, int B, lh_stream_t stream) {
    using namespace lh;
    if (!emb || !w || !bias || !ln_w || !ln_b || !scratch || !gain || B <= 0 || scratch == gain) return LH_ERR_ARG;
    hipLaunchKernelGGL(k_embed_proj, dim3((C * NF + EP_ROWS - 1) / EP_ROWS, (B + EP_NB - 1) / EP_NB), dim3(256), 0,
                       (hipStream_t)stream, emb, w, bias, scratch, B);
    hipLaunchKernelGGL(k_embed_ln, dim3(B), dim3(256), 0, (hipStream_t)stream, scratch, ln_w, ln_b, gain);
    return check_launch();
}

extern "C" int lh_abi_version(void) { return 14; }

extern "C" int lh_check_config(int nfft, int hop, int n_mics, int emb_dim, int n_blocks_unused, int lstm_hidden,
                               int n_heads, int attn_window, int n_srcs, int spk_emb_dim) {
    using namespace lh;
    (void)n_blocks_unused;
    if (nfft != NFFT || hop != HOP || n_mics != NMIC || emb_dim != C || lstm_hidden != H || n_heads != NH ||
        attn_window != WIN || n_srcs != NSRC || spk_emb_dim != SPK)
        return LH_ERR_UNSUPPORTED;
    return LH_OK;
}
