// Front end of the separator: STFT analysis + causal 3x3 conv, and the speaker-gain projection.
// HBM-streaming kernels (<2 % of the path's FLOPs): one workgroup owns a tile of
// frames, keeps samples and the spectrum tile in LDS, reads filter rows coalesced over frequency bins.
#include "lh_common.h"

namespace lh {

// ---- STFT analysis + causal 3x3 conv on fp32 MFMA (exact fp32), persistent workgroups --------------------------
// A tile = 14 output frames of one utterance (+2 halo frames of the causal conv = 16 STFT frames = one MFMA row
// tile per microphone).  The analysis filterbank lives in VGPRs for the whole kernel as B fragments (wave w owns
// filter-row tiles w, w+4, w+8(, 12): 144-192 registers), frames are staged once in LDS as the A image, the
// spectrum tile [4 ch][16 frames][97 bins] stays in LDS, and the conv is a [97 x 36] x [36 x 64] MFMA contraction
// per frame whose A operand is gathered straight from the spectrum tile; output rows leave through LDS as
// 256-byte coalesced stores.
constexpr int FE_TT = 14;                      // output frames per tile
constexpr int FE_NJ = 16;                      // STFT frames per tile (2 halo + 14)
constexpr int FE_KC = NFFT / 4;                // 48: k-chunk of one 16-lane group
constexpr int FE_KP = FE_KC + 4;               // 52: padded chunk row (13 x 16 B: conflict-free ds_read_b128)
constexpr int FE_SROW = NF + 3;                // spectrum row: [0]=0 pad, [1..97]=bins, [98]=0 pad, [99] unused
constexpr int FE_NT = (NK + 15) / 16;          // 13 filter-row tiles (194 -> 208)
constexpr int FE_OP = C + 4;                   // output staging row

__global__ void __launch_bounds__(256, 1) k_stft_conv_in(const float* __restrict__ x, const float* __restrict__ cbuf_in,
                                                          float* __restrict__ cbuf_out, const float* __restrict__ wfb_pk,
                                                          const float* __restrict__ wc_pk, const float* __restrict__ bc,
                                                          float* __restrict__ z, int B, int T, int n_samples) {
    __shared__ __attribute__((aligned(16))) float aimg[NMIC * 4 * FE_NJ * FE_KP];
    __shared__ float spec[2 * NMIC][FE_NJ][FE_SROW];
    __shared__ __attribute__((aligned(16))) float outs[NF * FE_OP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    // resident B fragments: filterbank tiles of this wave, conv weights of its 16 output channels
    float wf[4][FE_KC];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nt = min(wave + 4 * i, FE_NT - 1);
#pragma unroll
        for (int ks = 0; ks < FE_KC; ++ks) wf[i][ks] = wfb_pk[((long)nt * FE_KC + ks) * 64 + lane];
    }
    float wcv[9];
    int aoff[9];                               // spectrum-tile offset of conv tap q = g4*9 + ks -> (ch, kt, kf)
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
        wcv[ks] = wc_pk[(wave * 9 + ks) * 64 + lane];
        const int qq = g4 * 9 + ks, ch = qq / 9, kt = (qq % 9) / 3, kf = qq % 3;
        aoff[ks] = (ch * FE_NJ + kt) * FE_SROW + kf;
    }
    const float cbias = bc[wave * 16 + l15];

    for (int i = tid; i < 2 * NMIC * FE_NJ; i += 256) {        // zero the frequency padding columns once
        float* row = &spec[0][0][0] + i * FE_SROW;
        row[0] = 0.0f; row[NF + 1] = 0.0f; row[NF + 2] = 0.0f;
    }

    const int tiles_per_b = (T + FE_TT - 1) / FE_TT;
    for (int tile = blockIdx.x; tile < B * tiles_per_b; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile % tiles_per_b) * FE_TT;
        const int nt_out = min(FE_TT, T - t0);
        __syncthreads();                       // previous tile fully consumed (aimg / spec / outs)

        // stage the 16 frames of both microphones: frame j = samples (t0-2+j)*128 .. +192, 48 float4 each
        {
            float4 stg[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int e = tid + 256 * i;
                const int m = e / (FE_NJ * 48), j = (e / 48) % FE_NJ, c4 = e % 48;
                const int t = t0 - 2 + j;
                stg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t >= 0 && t < T)
                    stg[i] = *reinterpret_cast<const float4*>(&x[((long)b * NMIC + m) * n_samples + (long)t * HOP + c4 * 4]);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int e = tid + 256 * i;
                const int m = e / (FE_NJ * 48), j = (e / 48) % FE_NJ, c4 = e % 48;
                *reinterpret_cast<float4*>(&aimg[((m * 4 + c4 / 12) * FE_NJ + j) * FE_KP + (c4 % 12) * 4]) = stg[i];
            }
        }
        __syncthreads();

        // spectrum: [16 frames x 192] x [192 x 16 filter rows] per (mic, tile)
#pragma unroll
        for (int m = 0; m < NMIC; ++m) {
            float av[FE_KC];
            const float* arow = &aimg[((m * 4 + g4) * FE_NJ + l15) * FE_KP];
#pragma unroll
            for (int qq = 0; qq < FE_KC / 4; ++qq) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
                av[qq * 4 + 0] = a4.x; av[qq * 4 + 1] = a4.y; av[qq * 4 + 2] = a4.z; av[qq * 4 + 3] = a4.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nt = wave + 4 * i;
                if (nt < FE_NT) {              // wave-uniform
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < FE_KC; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wf[i][ks], acc, 0, 0, 0);
                    const int k = nt * 16 + l15;                       // filter row: k < 97 re, 97..193 im
                    if (k < NK) {
                        const int ch = (k / NF) * NMIC + m, f = k % NF;    // channels re_m0, re_m1, im_m0, im_m1
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = g4 * 4 + r, t = t0 - 2 + j;
                            float v = acc[r];
                            if (t < 0) v = cbuf_in[(((long)b * 4 + ch) * 2 + (t + 2)) * NF + f];   // carried halo frames
                            if (t >= T) v = 0.0f;
                            spec[ch][j][1 + f] = v;
                        }
                    }
                }
            }
        }
        __syncthreads();

        // new halo state = last two frames of the halo-extended spectrum (only the last tile holds them)
        if (t0 + nt_out == T) {
            for (int i = tid; i < 4 * 2 * NF; i += 256) {
                const int f = i % NF, r = (i / NF) % 2, ch = i / (2 * NF);
                cbuf_out[(((long)b * 4 + ch) * 2 + r) * NF + f] = spec[ch][(T - 2 + r) - t0 + 2][1 + f];
            }
        }

        // conv: per output frame [97 bins x 36 taps] x [36 x 16 channels of this wave]; A gathered from the tile
        for (int jt = 0; jt < nt_out; ++jt) {
            f32x4 acc[7];
#pragma unroll
            for (int mt = 0; mt < 7; ++mt) acc[mt] = f32x4{cbias, cbias, cbias, cbias};
            const float* sp = &spec[0][0][0] + jt * FE_SROW;
#pragma unroll
            for (int ks = 0; ks < 9; ++ks) {
#pragma unroll
                for (int mt = 0; mt < 7; ++mt) {
                    const int f = min(mt * 16 + l15, NF - 1);          // rows >= 97 are dropped below
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sp[aoff[ks] + f], wcv[ks], acc[mt], 0, 0, 0);
                }
            }
#pragma unroll
            for (int mt = 0; mt < 7; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = mt * 16 + g4 * 4 + r;
                    if (f < NF) outs[f * FE_OP + wave * 16 + l15] = acc[mt][r];
                }
            __syncthreads();
            float* dst = z + (((long)b * T + t0 + jt) * NF) * C;
            for (int e = tid; e < NF * 16; e += 256)
                *reinterpret_cast<float4*>(&dst[e * 4]) = *reinterpret_cast<const float4*>(&outs[(e >> 4) * FE_OP + (e & 15) * 4]);
            __syncthreads();
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.y * EP_NB;
    for (int i = tid; i < EP_NB * SPK; i += 256) {
        const int bb = min(b0 + i / SPK, B - 1);
        es[i / SPK][i % SPK] = emb[(long)bb * SPK + (i % SPK)];
    }
    __syncthreads();
    for (int rr = wave; rr < EP_ROWS; rr += 4) {
        const int r = blockIdx.x * EP_ROWS + rr;
        if (r >= N) break;
        const float4 w4 = *reinterpret_cast<const float4*>(&w[(long)r * SPK + lane * 4]);
        const float bz = bias[r];
#pragma unroll
        for (int j = 0; j < EP_NB; ++j) {
            const float4 e4 = *reinterpret_cast<const float4*>(&es[j][lane * 4]);
            float s = wave_sum(w4.x * e4.x + w4.y * e4.y + w4.z * e4.z + w4.w * e4.w);
            if (lane == 0 && b0 + j < B) raw[(long)(b0 + j) * N + r] = s + bz;
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
    for (int oidx = tid; oidx < N; oidx += 256) {
        const int f = oidx / C, c = oidx % C;
        const int i = c * NF + f;                     // reference flat order is channel-major (reshape [B,C,F])
        gain[(long)b * N + oidx] = (vals[i] - mean) * rstd * lnw[i] + lnb[i];
    }
}

}  // namespace lh

extern "C" int lh_stft_conv_in(const float* x, const float* conv_buf_in, float* conv_buf_out, const float* wfb_t,
                               const float* wconv_pk, const float* bconv, float* z, int B, int T, int n_samples,
                               lh_stream_t stream) {
    using namespace lh;
    if (!x || !conv_buf_in || !conv_buf_out || !wfb_t || !wconv_pk || !bconv || !z || B <= 0 || T <= 0) return LH_ERR_ARG;
    if (conv_buf_in == conv_buf_out || n_samples != T * HOP + (NFFT - HOP)) return LH_ERR_ARG;
    const int tiles = B * ((T + FE_TT - 1) / FE_TT);
    hipLaunchKernelGGL(k_stft_conv_in, dim3(tiles < 256 ? tiles : 256), dim3(256), 0, (hipStream_t)stream, x,
                       conv_buf_in, conv_buf_out, wfb_t, wconv_pk, bconv, z, B, T, n_samples);
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

extern "C" int lh_abi_version(void) { return 9; }

extern "C" int lh_check_config(int nfft, int hop, int n_mics, int emb_dim, int n_blocks_unused, int lstm_hidden,
                               int n_heads, int attn_window, int n_srcs, int spk_emb_dim) {
    using namespace lh;
    (void)n_blocks_unused;
    if (nfft != NFFT || hop != HOP || n_mics != NMIC || emb_dim != C || lstm_hidden != H || n_heads != NH ||
        attn_window != WIN || n_srcs != NSRC || spk_emb_dim != SPK)
        return LH_ERR_UNSUPPORTED;
    return LH_OK;
}
