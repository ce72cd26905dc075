// Front end of the separator: STFT analysis + causal 3x3 conv, and the speaker-gain projection.
// HBM-streaming kernels (<2 % of the path's FLOPs): one workgroup owns a tile of
// frames, keeps samples and the spectrum tile in LDS, reads filter rows coalesced over frequency bins.
#include "lh_common.h"

namespace lh {

constexpr int FE_TT = 8;                       // frames per workgroup
constexpr int FE_NJ = FE_TT + 2;               // + 2 halo frames of the causal conv
constexpr int FE_NS = FE_NJ * HOP + (NFFT - HOP);   // samples staged per mic
constexpr int FE_SROW = NF + 3;                // spectrum row: [0]=0 pad, [1..97]=bins, [98]=0 pad, [99] unused

// grid (ceil(T/8), B), block 256
__global__ void __launch_bounds__(256) k_stft_conv_in(const float* __restrict__ x, const float* __restrict__ cbuf_in,
                                                       float* __restrict__ cbuf_out, const float* __restrict__ wfb_t,
                                                       const float* __restrict__ wc_pk, const float* __restrict__ bc,
                                                       float* __restrict__ z, int T, int n_samples) {
    __shared__ float xs[NMIC][FE_NS];
    __shared__ float spec[2 * NMIC][FE_NJ][FE_SROW];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * FE_TT;
    const int nt = min(FE_TT, T - t0);

    // stage samples of frames t0-2 .. t0+TT-1 (zeros outside the signal)
    const long sb = (long)(t0 - 2) * HOP;
    for (int i = tid; i < NMIC * FE_NS; i += 256) {
        int m = i / FE_NS, s = i % FE_NS;
        long g = sb + s;
        xs[m][s] = (g >= 0 && g < n_samples) ? x[((long)b * NMIC + m) * n_samples + g] : 0.0f;
    }
    for (int i = tid; i < 2 * NMIC * FE_NJ; i += 256) {
        float* row = &spec[0][0][0] + i * FE_SROW;
        row[0] = 0.0f; row[NF + 1] = 0.0f; row[NF + 2] = 0.0f;
    }
    __syncthreads();

    // spectrum: thread k owns filter row k (k<97 real part of bin k, k>=97 imaginary part of bin k-97)
    if (tid < NK) {
        const int k = tid;
        float acc[NMIC][FE_NJ];
#pragma unroll
        for (int m = 0; m < NMIC; ++m)
#pragma unroll
            for (int j = 0; j < FE_NJ; ++j) acc[m][j] = 0.0f;
#pragma unroll 8
        for (int n = 0; n < NFFT; ++n) {
            const float w = wfb_t[n * NK + k];
#pragma unroll
            for (int m = 0; m < NMIC; ++m)
#pragma unroll
                for (int j = 0; j < FE_NJ; ++j) acc[m][j] = fmaf(xs[m][j * HOP + n], w, acc[m][j]);
        }
        const int f = k % NF;
        const int part = k / NF;                      // 0 = re, 1 = im
#pragma unroll
        for (int m = 0; m < NMIC; ++m) {
            const int ch = part * NMIC + m;           // channel order re_m0, re_m1, im_m0, im_m1
#pragma unroll
            for (int j = 0; j < FE_NJ; ++j) {
                const int t = t0 - 2 + j;
                float v = acc[m][j];
                if (t < 0) v = cbuf_in[(((long)b * 4 + ch) * 2 + (t + 2)) * NF + f];   // carried halo frames
                if (t >= T) v = 0.0f;
                spec[ch][j][1 + f] = v;
            }
        }
    }
    __syncthreads();

    // new halo state = last two frames of the halo-extended spectrum (only the last tile holds them)
    if (t0 + nt == T) {
        for (int i = tid; i < 4 * 2 * NF; i += 256) {
            int f = i % NF, r = (i / NF) % 2, ch = i / (2 * NF);
            int j = (T - 2 + r) - t0 + 2;
            cbuf_out[(((long)b * 4 + ch) * 2 + r) * NF + f] = spec[ch][j][1 + f];
        }
    }

    // 3x3 conv, 4 -> 64 channels: lane = output channel, the 4 waves stride over (frame, bin) positions
    const int o = tid & 63;
    const int grp = tid >> 6;
    float w[36];
#pragma unroll
    for (int q = 0; q < 36; ++q) w[q] = wc_pk[q * C + o];
    const float bias = bc[o];
    for (int pos = grp; pos < nt * NF; pos += 4) {
        const int jt = pos / NF, f = pos % NF;
        float a = bias;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int kf = 0; kf < 3; ++kf) a = fmaf(spec[ch][jt + kt][f + kf], w[(ch * 3 + kt) * 3 + kf], a);
        z[(((long)b * T + t0 + jt) * NF + f) * C + o] = a;
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
    hipLaunchKernelGGL(k_stft_conv_in, dim3((T + FE_TT - 1) / FE_TT, B), dim3(256), 0, (hipStream_t)stream, x,
                       conv_buf_in, conv_buf_out, wfb_t, wconv_pk, bconv, z, T, n_samples);
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

extern "C" int lh_abi_version(void) { return 3; }

extern "C" int lh_check_config(int nfft, int hop, int n_mics, int emb_dim, int n_blocks_unused, int lstm_hidden,
                               int n_heads, int attn_window, int n_srcs, int spk_emb_dim) {
    using namespace lh;
    (void)n_blocks_unused;
    if (nfft != NFFT || hop != HOP || n_mics != NMIC || emb_dim != C || lstm_hidden != H || n_heads != NH ||
        attn_window != WIN || n_srcs != NSRC || spk_emb_dim != SPK)
        return LH_ERR_UNSUPPORTED;
    return LH_OK;
}
