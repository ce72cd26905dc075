// Back end of the separator: causal ConvTranspose2d(64->4, 3x3) + re/im re-pack + iSTFT synthesis with
// overlap-add and carried tails, persistent workgroups of 8 waves.
//
// The transposed conv is factored so that every input frame is read from HBM once and multiplied once:
//     P[fr][f'][(kt,kf,o)] = sum_c Y[fr][f'][c] * Wd[c][o][kt][kf]          ([97 x 64] x [64 x 36] per frame)
//     D[t][o][f]           = b[o] + sum_{kt,kf} P[t-kt][f+1-kf][(kt,kf,o)]   (9-term gather-add from a 3-frame ring)
// A tile = 15 output frames of one utterance: 18 input frames (3 halo) stream through a register-staged LDS image,
// the 16 spectra Sx[t0..t0+15] (Sx[0] = carried istft_buf) are assembled in LDS as a second A image, and the
// synthesis filterbank ([194 x 192], resident in VGPRs as B fragments for the whole kernel) turns them into 16 frames
// per source that are overlap-added into 15 x 128 output samples.
// Both contractions run on split-precision fp16 MFMA (v = hi + 2^-11 lo, three v_mfma_f32_16x16x32_f16 per product,
// lh_split.h): the exact-fp32 MFMA version of this kernel spent 5x the matrix cycles and, at one wave per SIMD
// (386 registers), could not hide them: 0.46 ms per call at B = 32 against an HBM floor of 0.08 ms.
#include "lh_split.h"

namespace lh {

constexpr int BE_NT = 512;                // threads per workgroup (8 waves, two per SIMD)
constexpr int BE_TT = 15;                 // output frames per tile
constexpr int BE_NJ = BE_TT + 1;          // Sx frames t0 .. t0+15 (one MFMA row tile per source)
constexpr int BE_NP = 36;                 // partial-product columns (kt,kf,o); the B image pads them to 48
constexpr int BE_PP = BE_NP + 1;          // P row stride (odd: conflict-free gather)
constexpr int BE_SK = 7;                  // synthesis k-steps: 194 spectrum rows -> 224
constexpr int BE_SA = BE_SK * 4 * BE_NJ * 8;   // halves per source in the Sx A image
constexpr int BE_FP = NFFT + 4;           // synthesis frame staging row
constexpr int BE_NLD = (NF * 16 + BE_NT - 1) / BE_NT;   // float4 per thread and frame (4)
constexpr int BE_RING = 4;                // input frames in flight per workgroup

// grid = persistent (<= 256), block 512
__global__ void __launch_bounds__(BE_NT) k_deconv_istft(const float* __restrict__ y, const float* __restrict__ dbuf_in,
                                                        float* __restrict__ dbuf_out, const float* __restrict__ ibuf_in,
                                                        float* __restrict__ ibuf_out, const _Float16* __restrict__ wd_pk,
                                                        const float* __restrict__ bd, const _Float16* __restrict__ wfb_pk,
                                                        float* __restrict__ wave_out, int B, int T) {
    // Two input frames per loop iteration (half the barriers; at one frame the loop was ~80 % stall): two A images, and
    // a partial-product ring of 4 frames (2 being written while the gather still reads the 2 before them).
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * FR_A];                 // A images of two input frames
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * FR_A];
    // partial products of 4 frames; bin f sits in row f + 1, rows 0 and 98 stay zero (the f -/+ 1 taps of the edge bins)
    __shared__ float pring[4][NF + 2][BE_PP];
    __shared__ __attribute__((aligned(16))) _Float16 sxh[NSRC * BE_SA];             // A image of the 16 spectra
    __shared__ __attribute__((aligned(16))) _Float16 sxl[NSRC * BE_SA];
    // synthesis frames: only live after the frame loop, in the space of the (then dead) hi A images
    static_assert(sizeof(float) * BE_NJ * NSRC * BE_FP <= sizeof(_Float16) * 2 * FR_A, "frs must fit in ahi");
    float (*frs)[NSRC][BE_FP] = reinterpret_cast<float (*)[NSRC][BE_FP]>(ahi);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    // B fragments: the deconv taps [64 -> 48 columns] (3 column tiles x 2 k-steps, 6 KB) sit in LDS in fragment order;
    // the synthesis filterbank tiles 2w, 2w+1 (of 12; 112 registers) are fetched from L2 once per tile, right before
    // they are used — the registers hold the frame prefetch ring while the frames stream
    __shared__ __attribute__((aligned(16))) _Float16 wds[3 * 2 * 64 * 16];
    for (int i = tid; i < 3 * 2 * 64 * 2; i += BE_NT)
        *reinterpret_cast<f16x8*>(&wds[i * 8]) = *reinterpret_cast<const f16x8*>(&wd_pk[i * 8]);
    const bool synth = wave < 6;
    float bias4[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) bias4[o] = bd[o];

    // rows / k-padding of the A images that are never written only feed dropped outputs or multiply zero weights,
    // but must be finite
    for (int i = tid; i < 2 * FR_A; i += BE_NT) { ahi[i] = (_Float16)0.f; alo[i] = (_Float16)0.f; }
    for (int i = tid; i < NSRC * BE_SA; i += BE_NT) { sxh[i] = (_Float16)0.f; sxl[i] = (_Float16)0.f; }
    for (int i = tid; i < 4 * 2 * BE_PP; i += BE_NT) pring[i / (2 * BE_PP)][((i / BE_PP) & 1) * (NF + 1)][i % BE_PP] = 0.f;

    const int tiles_per_b = (T + BE_TT - 1) / BE_TT;
    const long L = (long)HOP * T;
    for (int tile = blockIdx.x; tile < B * tiles_per_b; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile % tiles_per_b) * BE_TT;
        const int nt_out = min(BE_TT, T - t0);
        __syncthreads();

        // one spectrum value -> row jd of source s's A image
        auto put_sx = [&](int jd, int s, int k, float v) {
            const _Float16 h = (_Float16)v;
            const int idx = s * BE_SA + a_index<BE_NJ>(jd, k);
            sxh[idx] = h;
            sxl[idx] = (_Float16)((v - (float)h) * PW_SPLIT);
        };
        // Sx frame 0 of the very first tile is the carried spectrum of the previous call
        if (t0 == 0)
            for (int i = tid; i < NSRC * NK; i += BE_NT) put_sx(0, i / NK, i % NK, ibuf_in[(long)b * NSRC * NK + i]);

        // input frames t0-3 .. t0+nt_out-1 ; after frame fr has been multiplied, output frame td = fr is complete
        float4 stg[BE_RING][BE_NLD];
        auto load_frame = [&](int fr, float4 (&dst)[BE_NLD]) {
#pragma unroll
            for (int i = 0; i < BE_NLD; ++i) {
                const int e = min(tid + BE_NT * i, NF * 16 - 1), f = e >> 4, c4 = e & 15;
                if (fr >= 0) {
                    dst[i] = *reinterpret_cast<const float4*>(&y[(((long)b * T + fr) * NF + f) * C + c4 * 4]);
                } else {                              // carried halo frames, layout [B][64][2][97]
                    const float* d0 = &dbuf_in[(((long)b * C + c4 * 4) * 2 + (fr + 2)) * NF + f];
                    dst[i] = make_float4(d0[0], d0[2 * NF], d0[4 * NF], d0[6 * NF]);
                }
            }
        };
        const int fr_first = max(t0 - 3, -2);         // frames below -2 do not exist (their taps see nothing)
        const int fr_end = t0 + nt_out;
        // BE_RING frames in flight per workgroup: with a single one the loop ran at one HBM round trip per frame
#pragma unroll
        for (int u = 0; u < BE_RING; ++u)
            if (fr_first + u < fr_end) load_frame(fr_first + u, stg[u]);
        for (int fbase = fr_first; fbase < fr_end; fbase += BE_RING) {
#pragma unroll
          for (int u = 0; u < BE_RING; u += 2) {
            const int fr = fbase + u;                 // this iteration: frames fr and fr + 1 (the second may not exist)
            if (fr >= fr_end) break;
            const bool two = fr + 1 < fr_end;
            // stage the frames into the two A images, refill their ring slots
#pragma unroll
            for (int i = 0; i < BE_NLD; ++i) {
                const int e = tid + BE_NT * i;
                if (e < NF * 16) {
                    store_split4<FR_RP>(ahi, alo, e >> 4, (e & 15) * 4, stg[u][i]);
                    if (two) store_split4<FR_RP>(ahi + FR_A, alo + FR_A, e >> 4, (e & 15) * 4, stg[u + 1][i]);
                }
            }
            __syncthreads();
            if (fr + BE_RING < fr_end) load_frame(fr + BE_RING, stg[u]);
            if (fr + 1 + BE_RING < fr_end) load_frame(fr + 1 + BE_RING, stg[u + 1]);

            // P[fr + q] = Y[fr + q] (97 x 64) * Wd (64 x 48): 2 x 21 (row tile, column tile) products over the 8 waves
            for (int p = wave; p < (two ? 42 : 21); p += 8) {
                const int q = p >= 21, pp = p - 21 * q, mt = pp / 3, nt = pp % 3;
                const int slot = (fr + q + 4) & 3;    // fr >= -2
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    wh[ks] = *reinterpret_cast<const f16x8*>(&wds[((nt * 2 + ks) * 64 + lane) * 16]);
                    wl[ks] = *reinterpret_cast<const f16x8*>(&wds[((nt * 2 + ks) * 64 + lane) * 16 + 8]);
                }
                const f32x4 acc = mma_tile<FR_RP, 2>(ahi + q * FR_A, alo + q * FR_A, mt, g4, l15, wh, wl, 0.f);
                const int col = nt * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = mt * 16 + g4 * 4 + r;
                    if (f < NF && col < BE_NP) pring[slot][f + 1][col] = acc[r];
                }
            }
            __syncthreads();

            // output frames td = fr, fr + 1: D[o][f] = b[o] + sum_{kt,kf} P[td-kt][f+1-kf][(kt*3+kf)*4 + o]
            for (int i = tid; i < (two ? 2 : 1) * 4 * NF; i += BE_NT) {
                const int q = i >= 4 * NF, ii = i - 4 * NF * q;
                const int td = fr + q;
                if (td < t0 - 1 || td < 0) continue;
                const int jd = td + 1 - t0;           // Sx frame index inside the tile
                const int o = ii & 3, f = ii >> 2;
                float v = bias4[o];
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    const int ps = (td - kt + 4) & 3;         // td >= 0, so frame td - kt >= -2 exists (halo or zero state)
#pragma unroll
                    for (int kf = 0; kf < 3; ++kf) v += pring[ps][f + 2 - kf][(kt * 3 + kf) * 4 + o];   // guard rows: no bounds
                }
                const int s = o >> 1, k = (o & 1) * NF + f;
                put_sx(jd, s, k, v);
                if (td == T - 1) ibuf_out[((long)b * NSRC + s) * NK + k] = v;      // new carried spectrum (exact fp32)
            }
            // the next iteration's staging barrier orders these pring reads before their slots are overwritten
          }
        }
        __syncthreads();

        // new carried conv halo (last tile only)
        if (t0 + nt_out == T) {
            for (int i = tid; i < 2 * NF * C; i += BE_NT) {
                const int c = i % C, f = (i / C) % NF, r = i / (C * NF);
                const int fr = T - 2 + r;
                const float v = fr >= 0 ? y[(((long)b * T + fr) * NF + f) * C + c]
                                        : dbuf_in[(((long)b * C + c) * 2 + (fr + 2)) * NF + f];
                dbuf_out[(((long)b * C + c) * 2 + r) * NF + f] = v;
            }
        }

        // synthesis: fr[jd][s][n] = sum_k Sx[jd][s][k] Wdec[k][n]; row tile = the 16 frames of a source, waves 0..5
        // own 2 of the 12 column tiles of 16 samples each
        if (synth) {
#pragma unroll 1
            for (int i = 0; i < 2; ++i) {             // one column tile at a time: 56 registers of B fragments
                f16x8 wfh[BE_SK], wfl[BE_SK];
                load_w<BE_SK>(wfb_pk, 2 * wave + i, lane, wfh, wfl);
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const f32x4 acc = mma_tile<BE_NJ, BE_SK>(sxh + s * BE_SA, sxl + s * BE_SA, 0, g4, l15, wfh, wfl, 0.f);
#pragma unroll
                    for (int r = 0; r < 4; ++r) frs[g4 * 4 + r][s][(2 * wave + i) * 16 + l15] = acc[r];
                }
            }
        }
        __syncthreads();

        // overlap-add: output frame t (samples 128t..128t+127) = fr[t+1][0:128] + fr[t][128:192]
        for (int i = tid; i < nt_out * NSRC * HOP; i += BE_NT) {
            const int n = i % HOP, s = (i / HOP) % NSRC, jt = i / (HOP * NSRC);
            float v = frs[jt + 1][s][n];
            if (n < NFFT - HOP) v += frs[jt][s][n + HOP];
            wave_out[((long)b * NSRC + s) * L + (long)(t0 + jt) * HOP + n] = v;
        }
        __syncthreads();
        // frs lived in the hi A images: their pad rows (97..111 feed dropped outputs) must hold finite numbers again
        for (int i = tid; i < 2 * FR_A; i += BE_NT) ahi[i] = (_Float16)0.f;
    }
}

}  // namespace lh

extern "C" int lh_deconv_istft(const float* y, const float* deconv_buf_in, float* deconv_buf_out,
                               const float* istft_buf_in, float* istft_buf_out, const void* wdec_pk,
                               const float* bdec, const void* wfb_dec, float* wave_out, int B, int T,
                               lh_stream_t stream) {
    using namespace lh;
    if (!y || !deconv_buf_in || !deconv_buf_out || !istft_buf_in || !istft_buf_out || !wdec_pk || !bdec || !wfb_dec ||
        !wave_out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (deconv_buf_in == deconv_buf_out || istft_buf_in == istft_buf_out) return LH_ERR_ARG;
    const int tiles = B * ((T + BE_TT - 1) / BE_TT);
    hipLaunchKernelGGL(k_deconv_istft, dim3(tiles < 256 ? tiles : 256), dim3(BE_NT), 0, (hipStream_t)stream, y,
                       deconv_buf_in, deconv_buf_out, istft_buf_in, istft_buf_out, (const _Float16*)wdec_pk, bdec,
                       (const _Float16*)wfb_dec, wave_out, B, T);
    return check_launch();
}
