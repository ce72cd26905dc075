// Back end of the separator: causal ConvTranspose2d(64->4, 3x3) + re/im re-pack + iSTFT synthesis with
// overlap-add and carried tails.  <1 % of the path's FLOPs; one
// workgroup owns 8 output frames (1024 samples per source), keeps the 9 spectra it needs in LDS and reads
// the synthesis filterbank rows coalesced over the sample index.
#include "lh_common.h"

namespace lh {

constexpr int BE_TT = 8;
constexpr int BE_NJ = BE_TT + 1;          // Sx frames t0 .. t0+TT  (Sx frame 0 = carried istft_buf, frame t'+1 = D[t'])
constexpr int BE_KP = NK + 2;             // 196

// grid (ceil(T/8), B), block 256
__global__ void __launch_bounds__(256) k_deconv_istft(const float* __restrict__ y, const float* __restrict__ dbuf_in,
                                                      float* __restrict__ dbuf_out, const float* __restrict__ ibuf_in,
                                                      float* __restrict__ ibuf_out, const float* __restrict__ wd_pk,
                                                      const float* __restrict__ bd, const float* __restrict__ wfb,
                                                      float* __restrict__ wave_out, int T) {
    __shared__ __attribute__((aligned(16))) float wd[4 * 9 * C];          // [o][kt][kf][c]
    __shared__ float sx[BE_NJ][NSRC][BE_KP];
    __shared__ float frs[BE_NJ][NSRC][NFFT];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * BE_TT;
    const int nt = min(BE_TT, T - t0);

    for (int i = tid; i < 4 * 9 * C; i += 256) wd[i] = wd_pk[i];
    __syncthreads();

    // ---- deconv: item = (Sx frame jd, bin f) -> the 4 output channels o = 2*src + {re,im}
    for (int item = tid; item < (nt + 1) * NF; item += 256) {
        const int jd = item / NF, f = item % NF;
        const int tp = t0 + jd;                       // Sx frame index
        float o4[4];
        if (tp == 0) {                                // carried last spectrum of the previous call
#pragma unroll
            for (int o = 0; o < 4; ++o) o4[o] = ibuf_in[((long)b * NSRC + (o >> 1)) * NK + (o & 1) * NF + f];
        } else {
            const int td = tp - 1;                    // deconv output frame
#pragma unroll
            for (int o = 0; o < 4; ++o) o4[o] = bd[o];
            for (int kt = 0; kt < 3; ++kt) {
                const int fr = td - kt;               // input frame (>= -2)
                for (int kf = 0; kf < 3; ++kf) {
                    const int fi = f + 1 - kf;
                    if (fi < 0 || fi >= NF) continue;
                    const float* w0 = &wd[((0 * 3 + kt) * 3 + kf) * C];
                    if (fr >= 0) {
                        const float* src = y + (((long)b * T + fr) * NF + fi) * C;
#pragma unroll
                        for (int c4 = 0; c4 < C / 4; ++c4) {
                            const float4 v = *reinterpret_cast<const float4*>(src + c4 * 4);
#pragma unroll
                            for (int o = 0; o < 4; ++o) {
                                const float4 w = *reinterpret_cast<const float4*>(w0 + o * 9 * C + c4 * 4);
                                o4[o] += v.x * w.x + v.y * w.y + v.z * w.z + v.w * w.w;
                            }
                        }
                    } else {                          // carried halo frames, layout [B][64][2][97]
                        for (int c = 0; c < C; ++c) {
                            const float v = dbuf_in[(((long)b * C + c) * 2 + (fr + 2)) * NF + fi];
#pragma unroll
                            for (int o = 0; o < 4; ++o) o4[o] += v * w0[o * 9 * C + c];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) sx[jd][o >> 1][(o & 1) * NF + f] = o4[o];
    }
    __syncthreads();

    // ---- new carried state (last tile only): last two halo-extended input frames, last spectrum
    if (t0 + nt == T) {
        for (int i = tid; i < 2 * NF * C; i += 256) {
            const int c = i % C, f = (i / C) % NF, r = i / (C * NF);
            const int fr = T - 2 + r;
            const float v = fr >= 0 ? y[(((long)b * T + fr) * NF + f) * C + c]
                                    : dbuf_in[(((long)b * C + c) * 2 + (fr + 2)) * NF + f];
            dbuf_out[(((long)b * C + c) * 2 + r) * NF + f] = v;
        }
        for (int i = tid; i < NSRC * NK; i += 256) ibuf_out[(long)b * NSRC * NK + i] = sx[nt][i / NK][i % NK];
    }

    // ---- synthesis frames fr[jd][s][n] = sum_k Sx[jd][s][k] * Wdec[k][n]; thread = sample index n
    if (tid < NFFT) {
        float acc[BE_NJ][NSRC];
#pragma unroll
        for (int j = 0; j < BE_NJ; ++j)
#pragma unroll
            for (int s = 0; s < NSRC; ++s) acc[j][s] = 0.f;
#pragma unroll 8
        for (int k = 0; k < NK; ++k) {
            const float w = wfb[k * NFFT + tid];
#pragma unroll
            for (int j = 0; j < BE_NJ; ++j)
#pragma unroll
                for (int s = 0; s < NSRC; ++s) acc[j][s] = fmaf(sx[j][s][k], w, acc[j][s]);
        }
#pragma unroll
        for (int j = 0; j < BE_NJ; ++j)
#pragma unroll
            for (int s = 0; s < NSRC; ++s) frs[j][s][tid] = acc[j][s];
    }
    __syncthreads();

    // ---- overlap-add: output frame t (samples 128t..128t+127) = fr[t+1][0:128] + fr[t][128:192]
    const long L = (long)HOP * T;
    for (int i = tid; i < nt * NSRC * HOP; i += 256) {
        const int n = i % HOP, s = (i / HOP) % NSRC, jt = i / (HOP * NSRC);
        float v = frs[jt + 1][s][n];
        if (n < NFFT - HOP) v += frs[jt][s][n + HOP];
        wave_out[((long)b * NSRC + s) * L + (long)(t0 + jt) * HOP + n] = v;
    }
}

}  // namespace lh

extern "C" int lh_deconv_istft(const float* y, const float* deconv_buf_in, float* deconv_buf_out,
                               const float* istft_buf_in, float* istft_buf_out, const float* wdec_pk,
                               const float* bdec, const float* wfb_dec, float* wave_out, int B, int T,
                               lh_stream_t stream) {
    using namespace lh;
    if (!y || !deconv_buf_in || !deconv_buf_out || !istft_buf_in || !istft_buf_out || !wdec_pk || !bdec || !wfb_dec ||
        !wave_out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (deconv_buf_in == deconv_buf_out || istft_buf_in == istft_buf_out) return LH_ERR_ARG;
    hipLaunchKernelGGL(k_deconv_istft, dim3((T + BE_TT - 1) / BE_TT, B), dim3(256), 0, (hipStream_t)stream, y,
                       deconv_buf_in, deconv_buf_out, istft_buf_in, istft_buf_out, wdec_pk, bdec, wfb_dec, wave_out, T);
    return check_launch();
}
