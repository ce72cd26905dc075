// Back end of the separator: causal ConvTranspose2d(64->4, 3x3) + re/im re-pack + iSTFT synthesis with
// overlap-add and carried tails on exact fp32 MFMA, persistent
// workgroups.
//
// The transposed conv is factored so that every input frame is read from HBM once and multiplied once:
//     P[fr][f'][(kt,kf,o)] = sum_c Y[fr][f'][c] * Wd[c][o][kt][kf]          ([97 x 64] x [64 x 36] MFMA per frame)
//     D[t][o][f]           = b[o] + sum_{kt,kf} P[t-kt][f+1-kf][(kt,kf,o)]   (9-term gather-add from a 3-frame ring)
// A tile = 15 output frames of one utterance: 18 input frames (3 halo) stream through a register-staged LDS
// image, the 16 spectra Sx[t0..t0+15] (Sx[0] = carried istft_buf) are assembled in LDS, and the synthesis
// filterbank ([194 x 192], resident in VGPRs as MFMA B fragments for the whole kernel) turns them into 16
// frames per source that are overlap-added into 15 x 128 output samples.
#include "lh_common.h"

namespace lh {

constexpr int BE_TT = 15;                 // output frames per tile
constexpr int BE_NJ = BE_TT + 1;          // Sx frames t0 .. t0+15 (one MFMA row tile per source)
constexpr int BE_YP = 20;                 // Y image: 16-float k-chunk + 4 pad
constexpr int BE_RP = 112;                // 7 row tiles cover 97 bins
constexpr int BE_NP = 48;                 // 36 partial-product columns (kt,kf,o) padded to 3 column tiles
constexpr int BE_PP = BE_NP + 1;          // P row stride (odd: conflict-free gather)
constexpr int BE_KC = 52;                 // synthesis k-chunk per 16-lane group (194 -> 208 = 4 x 52)
constexpr int BE_SP = BE_KC + 4;          // Sx image chunk row (14 x 16 B)
constexpr int BE_FP = NFFT + 4;           // synthesis frame staging row

// grid = persistent (<= 256), block 256
__global__ void __launch_bounds__(256, 1) k_deconv_istft(const float* __restrict__ y, const float* __restrict__ dbuf_in,
                                                         float* __restrict__ dbuf_out, const float* __restrict__ ibuf_in,
                                                         float* __restrict__ ibuf_out, const float* __restrict__ wd_pk,
                                                         const float* __restrict__ bd, const float* __restrict__ wfb_pk,
                                                         float* __restrict__ wave_out, int B, int T) {
    __shared__ __attribute__((aligned(16))) float yimg[4 * BE_RP * BE_YP];          // A image of one input frame
    __shared__ float pring[3][NF][BE_PP];                                           // partial products of 3 frames
    __shared__ __attribute__((aligned(16))) float sximg[NSRC * 4 * BE_NJ * BE_SP];  // A image of the 16 spectra
    __shared__ __attribute__((aligned(16))) float frs[BE_NJ][NSRC][BE_FP];          // synthesis frames
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    // resident B fragments: deconv taps [64 x 48] (3 column tiles), synthesis filterbank tiles w, w+4, w+8
    float wd[3][16];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wd[nt][ks] = wd_pk[(nt * 16 + ks) * 64 + lane];
    float wf[3][BE_KC];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int ks = 0; ks < BE_KC; ++ks) wf[i][ks] = wfb_pk[((long)(wave + 4 * i) * BE_KC + ks) * 64 + lane];
    float bias4[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bias4[o] = bd[o];

    // zero the rows / k-padding of the A images that are never written (they only feed dropped outputs, but must
    // be finite)
    for (int i = tid; i < 4 * BE_RP * BE_YP; i += 256) yimg[i] = 0.f;
    for (int i = tid; i < NSRC * 4 * BE_NJ * BE_SP; i += 256) sximg[i] = 0.f;

    const int tiles_per_b = (T + BE_TT - 1) / BE_TT;
    const long L = (long)HOP * T;
    for (int tile = blockIdx.x; tile < B * tiles_per_b; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile % tiles_per_b) * BE_TT;
        const int nt_out = min(BE_TT, T - t0);
        __syncthreads();

        // Sx frame 0 of the very first tile is the carried spectrum of the previous call
        if (t0 == 0)
            for (int i = tid; i < NSRC * NK; i += 256) {
                const int s = i / NK, k = i % NK;
                sximg[((s * 4 + k / BE_KC) * BE_NJ + 0) * BE_SP + k % BE_KC] = ibuf_in[(long)b * NSRC * NK + i];
            }

        // input frames t0-3 .. t0+nt_out-1 ; after frame fr has been multiplied, output frame td = fr is complete
        float4 stg[7];
        auto load_frame = [&](int fr) {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int e = min(tid + 256 * i, NF * 16 - 1), f = e >> 4, c4 = e & 15;
                if (fr >= 0) {
                    stg[i] = *reinterpret_cast<const float4*>(&y[(((long)b * T + fr) * NF + f) * C + c4 * 4]);
                } else {                              // carried halo frames, layout [B][64][2][97]
                    const float* d0 = &dbuf_in[(((long)b * C + c4 * 4) * 2 + (fr + 2)) * NF + f];
                    stg[i] = make_float4(d0[0], d0[2 * NF], d0[4 * NF], d0[6 * NF]);
                }
            }
        };
        const int fr_first = max(t0 - 3, -2);         // frames below -2 do not exist (their taps see nothing)
        load_frame(fr_first);
        for (int fr = fr_first; fr < t0 + nt_out; ++fr) {
            // stage frame fr into the A image, prefetch the next one
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int e = tid + 256 * i;
                if (e < NF * 16) {
                    const int f = e >> 4, c4 = e & 15;
                    *reinterpret_cast<float4*>(&yimg[((c4 >> 2) * BE_RP + f) * BE_YP + (c4 & 3) * 4]) = stg[i];
                }
            }
            __syncthreads();
            if (fr + 1 < t0 + nt_out) load_frame(fr + 1);

            // P[fr] = Y[fr] (97 x 64) * Wd (64 x 48): wave w takes row tiles w and w+4
            const int slot = ((fr % 3) + 3) % 3;
            for (int mt = wave; mt < 7; mt += 4) {
                f32x4 acc[3];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* arow = &yimg[(g4 * BE_RP + mt * 16 + l15) * BE_YP];
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
                    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nt = 0; nt < 3; ++nt)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wd[nt][qq * 4 + j], acc[nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = mt * 16 + g4 * 4 + r;
                        if (f < NF) pring[slot][f][nt * 16 + l15] = acc[nt][r];
                    }
            }
            __syncthreads();

            // output frame td = fr: D[o][f] = b[o] + sum_{kt,kf} P[td-kt][f+1-kf][(kt*3+kf)*4 + o]
            const int td = fr;
            if (td >= t0 - 1 && td >= 0) {
                const int jd = td + 1 - t0;           // Sx frame index inside the tile
                for (int i = tid; i < 4 * NF; i += 256) {
                    const int o = i & 3, f = i >> 2;
                    float v = bias4[o];
#pragma unroll
                    for (int kt = 0; kt < 3; ++kt) {
                        const int pf = td - kt;
                        if (pf < -2) continue;
                        const int ps = ((pf % 3) + 3) % 3;
#pragma unroll
                        for (int kf = 0; kf < 3; ++kf) {
                            const int fi = f + 1 - kf;
                            if (fi >= 0 && fi < NF) v += pring[ps][fi][(kt * 3 + kf) * 4 + o];
                        }
                    }
                    const int s = o >> 1, k = (o & 1) * NF + f;
                    sximg[((s * 4 + k / BE_KC) * BE_NJ + jd) * BE_SP + k % BE_KC] = v;
                }
            }
            // the next iteration's staging barrier orders these pring reads before the slot is overwritten 3 frames later
        }
        __syncthreads();

        // new carried state (last tile only)
        if (t0 + nt_out == T) {
            for (int i = tid; i < 2 * NF * C; i += 256) {
                const int c = i % C, f = (i / C) % NF, r = i / (C * NF);
                const int fr = T - 2 + r;
                const float v = fr >= 0 ? y[(((long)b * T + fr) * NF + f) * C + c]
                                        : dbuf_in[(((long)b * C + c) * 2 + (fr + 2)) * NF + f];
                dbuf_out[(((long)b * C + c) * 2 + r) * NF + f] = v;
            }
            for (int i = tid; i < NSRC * NK; i += 256) {
                const int s = i / NK, k = i % NK;
                ibuf_out[(long)b * NSRC * NK + i] = sximg[((s * 4 + k / BE_KC) * BE_NJ + nt_out) * BE_SP + k % BE_KC];
            }
        }

        // synthesis: fr[jd][s][n] = sum_k Sx[jd][s][k] Wdec[k][n]; row tile = source, 12 column tiles of 16 samples
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            float av[BE_KC];
            const float* arow = &sximg[((s * 4 + g4) * BE_NJ + l15) * BE_SP];
#pragma unroll
            for (int qq = 0; qq < BE_KC / 4; ++qq) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
                av[qq * 4 + 0] = a4.x; av[qq * 4 + 1] = a4.y; av[qq * 4 + 2] = a4.z; av[qq * 4 + 3] = a4.w;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < BE_KC; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wf[i][ks], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) frs[g4 * 4 + r][s][(wave + 4 * i) * 16 + l15] = acc[r];
            }
        }
        __syncthreads();

        // overlap-add: output frame t (samples 128t..128t+127) = fr[t+1][0:128] + fr[t][128:192]
        for (int i = tid; i < nt_out * NSRC * HOP; i += 256) {
            const int n = i % HOP, s = (i / HOP) % NSRC, jt = i / (HOP * NSRC);
            float v = frs[jt + 1][s][n];
            if (n < NFFT - HOP) v += frs[jt][s][n + HOP];
            wave_out[((long)b * NSRC + s) * L + (long)(t0 + jt) * HOP + n] = v;
        }
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
    const int tiles = B * ((T + BE_TT - 1) / BE_TT);
    hipLaunchKernelGGL(k_deconv_istft, dim3(tiles < 256 ? tiles : 256), dim3(256), 0, (hipStream_t)stream, y,
                       deconv_buf_in, deconv_buf_out, istft_buf_in, istft_buf_out, wdec_pk, bdec, wfb_dec, wave_out, B, T);
    return check_launch();
}
