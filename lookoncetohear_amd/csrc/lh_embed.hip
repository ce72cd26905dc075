// Enrollment embedder (reference src/models/tfgridnet_orig/tfgridnet.py:88-127 on top of the
// espnet2 TF-GridNet trunk, see oracle/embedder_oracle.py): kernels that have no counterpart in the separator path.
//   k_emb_std / k_emb_stft_conv / k_emb_gn_apply   std-normalise, STFT(128/64, hann, centred) + Conv2d 3x3 + GroupNorm
//   k_emb_gx / k_emb_lstm / k_emb_convt_res        LN + unfold(4) input GEMM, BiLSTM recurrence, ConvTranspose1d + res
//   k_emb_vt / k_gemm_nt / k_emb_softmax           full (T x T) attention as two split-precision GEMMs, head merge fused
//   k_emb_head / k_emb_head_mean                   Linear(65*64 -> 256) + LayerNorm + mean over frames
// Activations are channel-last [B][T][65][64] like the separator's.
#include "lh_common.h"
#include "lh_quad.h"

namespace lh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int EF = 65;            // n_fft/2 + 1
constexpr int EK = 2 * EF;        // 130 filter rows (re | im)
constexpr int ENFFT = 128;
constexpr int EHOP = 64;
constexpr int EKS = 4;            // emb_ks (unfold / ConvTranspose1d kernel)
constexpr int EE = 8;             // ceil(512 / 65)
constexpr int EDQK = EF * EE;     // 520
constexpr int EDV = EF * VD;      // 1040
// Split form of the embedder kernels: the separator's un-rescaled v = hi + lo (split_hl, lh_common.h; the matrix core takes
// fp16 subnormals at full value); rounds 1-2 stored lo * 2048 here.  ESPLIT = 1 remains in the accumulator recombinations
// (am + ac / ESPLIT), where it folds away at compile time.
constexpr float ESPLIT = 1.0f;

// ---------------------------------------------------------------------------------------------------------------
// 1 / std(x[b]) over all samples of all microphones, unbiased (torch.std default) — tfgridnet_orig/tfgridnet.py:109
// ---------------------------------------------------------------------------------------------------------------
// One workgroup of 16 waves per utterance, 16-byte loads, four independent fp64 accumulator pairs per thread.  (Rounds 1-5:
// 256 threads walking 625 dependent scalar load -> convert -> add rounds each: 254 us per call at B = 64 for 41 MB.)
constexpr int ES_NTH = 1024;
__global__ void __launch_bounds__(ES_NTH) k_emb_std(const float* __restrict__ x, float* __restrict__ inv_std, int n) {
    __shared__ double red[2][ES_NTH / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    const float* xb = x + (long)blockIdx.x * n;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    // ONE summation order whatever the row's alignment (rows of a batch alternate between aligned and not when
    // NMIC * n_samples is not a multiple of 4, and the same utterance must give the same bits at every batch position):
    // element i always goes to thread (i / 4) % ES_NTH, accumulator i % 4; an unaligned row only loads its quads as scalars.
    const bool vec = ((reinterpret_cast<unsigned long long>(xb) & 15ull) == 0ull);
    const int n4 = n / 4;
    for (int i = tid; i < n4; i += ES_NTH) {
        float4 v4;
        if (vec) v4 = *reinterpret_cast<const float4*>(&xb[i * 4]);
        else { v4.x = xb[i * 4]; v4.y = xb[i * 4 + 1]; v4.z = xb[i * 4 + 2]; v4.w = xb[i * 4 + 3]; }
        const double a = v4.x, b = v4.y, c = v4.z, d = v4.w;
        s[0] += a; ss[0] += a * a;
        s[1] += b; ss[1] += b * b;
        s[2] += c; ss[2] += c * c;
        s[3] += d; ss[3] += d * d;
    }
    for (int i = n4 * 4 + tid; i < n; i += ES_NTH) { const double v = xb[i]; s[0] += v; ss[0] += v * v; }
    double st = (s[0] + s[1]) + (s[2] + s[3]), sst = (ss[0] + ss[1]) + (ss[2] + ss[3]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o); sst += __shfl_xor(sst, o); }
    if (lane == 0) { red[0][wave] = st; red[1][wave] = sst; }
    __syncthreads();
    if (tid == 0) {
        double S = 0.0, SS = 0.0;
#pragma unroll
        for (int w = 0; w < ES_NTH / 64; ++w) { S += red[0][w]; SS += red[1][w]; }
        const double var = (SS - S * S / n) / (n - 1);
        inv_std[blockIdx.x] = (float)(1.0 / sqrt(var));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// STFT (n_fft 128, hop 64, periodic hann folded into the filter rows, centre = reflect padding) + re/im channel
// stacking + Conv2d(4 -> 64, 3x3, zero padding 1 in time AND frequency) on fp32 MFMA; also the per-tile partial
// sums of the GroupNorm(1, 64) that follows.  Same structure as k_stft_conv_in (lh_frontend.hip): 14 output frames
// per tile = 16 STFT frames (one halo frame each side), filterbank resident in VGPRs as B fragments.
// ---------------------------------------------------------------------------------------------------------------
constexpr int EM_TT = 14;
constexpr int EM_NJ = 16;
constexpr int EM_KC = ENFFT / 4;               // 32
constexpr int EM_KP = EM_KC + 4;               // 36 floats = 9 x 16 B
constexpr int EM_SROW = EF + 3;                // [0] pad, [1..65] bins, [66] pad, [67] unused
constexpr int EM_NT = (EK + 15) / 16;          // 9 filter-row tiles
constexpr int EM_OP = C + 4;

__global__ void __launch_bounds__(256, 1) k_emb_stft_conv(const float* __restrict__ x, const float* __restrict__ inv_std,
                                                           const float* __restrict__ wfb_pk, const float* __restrict__ wc_pk,
                                                           const float* __restrict__ bc, float* __restrict__ z,
                                                           double* __restrict__ gn_part, int B, int T, int n_samples) {
    __shared__ __attribute__((aligned(16))) float aimg[NMIC * 4 * EM_NJ * EM_KP];
    __shared__ float spec[2 * NMIC][EM_NJ][EM_SROW];
    __shared__ __attribute__((aligned(16))) float outs[EF * EM_OP];
    __shared__ double gred[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;

    float wf[3][EM_KC];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int nt = min(wave + 4 * i, EM_NT - 1);
#pragma unroll
        for (int ks = 0; ks < EM_KC; ++ks) wf[i][ks] = wfb_pk[((long)nt * EM_KC + ks) * 64 + lane];
    }
    float wcv[9];
    int aoff[9];
#pragma unroll
    for (int ks = 0; ks < 9; ++ks) {
        wcv[ks] = wc_pk[(wave * 9 + ks) * 64 + lane];
        const int qq = g4 * 9 + ks, ch = qq / 9, kt = (qq % 9) / 3, kf = qq % 3;
        aoff[ks] = (ch * EM_NJ + kt) * EM_SROW + kf;
    }
    const float cbias = bc[wave * 16 + l15];
    for (int i = tid; i < 2 * NMIC * EM_NJ; i += 256) {
        float* row = &spec[0][0][0] + i * EM_SROW;
        row[0] = 0.0f; row[EF + 1] = 0.0f; row[EF + 2] = 0.0f;
    }

    const int tiles_per_b = (T + EM_TT - 1) / EM_TT;
    for (int tile = blockIdx.x; tile < B * tiles_per_b; tile += gridDim.x) {
        const int b = tile / tiles_per_b;
        const int t0 = (tile % tiles_per_b) * EM_TT;
        const int nt_out = min(EM_TT, T - t0);
        const float sc = inv_std[b];
        __syncthreads();

        // frames t0-1 .. t0+14; frame t = padded samples t*64 .. +127, padded[i] = x[reflect(i - 64)]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const int m = e / (EM_NJ * 32), j = (e / 32) % EM_NJ, c4 = e % 32;
            const int t = t0 - 1 + j;
            const float* xb = x + ((long)b * NMIC + m) * n_samples;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0 && t < T) {
                const int i0 = t * EHOP + c4 * 4 - ENFFT / 2;
                if (i0 >= 0 && i0 + 3 < n_samples && ((n_samples | i0) & 3) == 0) {
                    v = *reinterpret_cast<const float4*>(&xb[i0]);
                } else {
                    float tmp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int idx = i0 + u;
                        if (idx < 0) idx = -idx;
                        if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
                        tmp[u] = xb[idx];
                    }
                    v = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
                }
                v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            }
            *reinterpret_cast<float4*>(&aimg[((m * 4 + c4 / 8) * EM_NJ + j) * EM_KP + (c4 % 8) * 4]) = v;
        }
        __syncthreads();

#pragma unroll
        for (int m = 0; m < NMIC; ++m) {
            float av[EM_KC];
            const float* arow = &aimg[((m * 4 + g4) * EM_NJ + l15) * EM_KP];
#pragma unroll
            for (int qq = 0; qq < EM_KC / 4; ++qq) {
                const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
                av[qq * 4 + 0] = a4.x; av[qq * 4 + 1] = a4.y; av[qq * 4 + 2] = a4.z; av[qq * 4 + 3] = a4.w;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int nt = wave + 4 * i;
                if (nt < EM_NT) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < EM_KC; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wf[i][ks], acc, 0, 0, 0);
                    const int k = nt * 16 + l15;
                    if (k < EK) {
                        const int ch = (k / EF) * NMIC + m, f = k % EF;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int j = g4 * 4 + r, t = t0 - 1 + j;
                            spec[ch][j][1 + f] = (t >= 0 && t < T) ? acc[r] : 0.0f;      // zero padding in time
                        }
                    }
                }
            }
        }
        __syncthreads();

        float gs = 0.f, gss = 0.f;
        for (int jt = 0; jt < nt_out; ++jt) {
            f32x4 acc[5];
#pragma unroll
            for (int mt = 0; mt < 5; ++mt) acc[mt] = f32x4{cbias, cbias, cbias, cbias};
            const float* sp = &spec[0][0][0] + jt * EM_SROW;
#pragma unroll
            for (int ks = 0; ks < 9; ++ks)
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) {
                    const int f = min(mt * 16 + l15, EF - 1);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sp[aoff[ks] + f], wcv[ks], acc[mt], 0, 0, 0);
                }
#pragma unroll
            for (int mt = 0; mt < 5; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = mt * 16 + g4 * 4 + r;
                    if (f < EF) { outs[f * EM_OP + wave * 16 + l15] = acc[mt][r]; gs += acc[mt][r]; gss += acc[mt][r] * acc[mt][r]; }
                }
            __syncthreads();
            float* dst = z + (((long)b * T + t0 + jt) * EF) * C;
            for (int e = tid; e < EF * 16; e += 256)
                *reinterpret_cast<float4*>(&dst[e * 4]) = *reinterpret_cast<const float4*>(&outs[(e >> 4) * EM_OP + (e & 15) * 4]);
            __syncthreads();
        }
        // GroupNorm partial sums of this tile (fp64)
        double ds = gs, dss = gss;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ds += __shfl_xor(ds, o); dss += __shfl_xor(dss, o); }
        if (lane == 0) { gred[0][wave] = ds; gred[1][wave] = dss; }
        __syncthreads();
        if (tid == 0) {
            gn_part[(long)tile * 2 + 0] = gred[0][0] + gred[0][1] + gred[0][2] + gred[0][3];
            gn_part[(long)tile * 2 + 1] = gred[1][0] + gred[1][1] + gred[1][2] + gred[1][3];
        }
    }
}

// GroupNorm(1, 64) over (C, T, F) of one utterance, affine per channel; in place.  grid (chunks, B)
// one float4 (channels 4q .. 4q+3) of a 64-channel row held by 16 consecutive lanes; every lane of the group must call it
__device__ __forceinline__ void ln_split_store(float4 u, _Float16* __restrict__ xh, _Float16* __restrict__ xl, long off, bool valid) {
    const float mean = group16_sum(u.x + u.y + u.z + u.w) * (1.0f / C);
    u.x -= mean; u.y -= mean; u.z -= mean; u.w -= mean;
    const float var = group16_sum(u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w) * (1.0f / C);
    const float rstd = rsqrtf(var + LN_EPS);
    const float v[4] = {u.x * rstd, u.y * rstd, u.z * rstd, u.w * rstd};
    f16x4 h4, l4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        _Float16 h, l;
        split_hl(v[i], h, l);
        h4[i] = h;
        l4[i] = l;
    }
    if (valid) {
        *reinterpret_cast<f16x4*>(&xh[off]) = h4;
        *reinterpret_cast<f16x4*>(&xl[off]) = l4;
    }
}
__global__ void __launch_bounds__(256) k_emb_gn_apply(float* __restrict__ z, const double* __restrict__ gn_part,
                                                      const float* __restrict__ gw, const float* __restrict__ gb,
                                                      _Float16* __restrict__ xs_next, long rows_x, int T) {
    __shared__ float stat[2];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int tiles_per_b = (T + EM_TT - 1) / EM_TT;
    if (tid == 0) {
        double s = 0.0, ss = 0.0;
        for (int i = 0; i < tiles_per_b; ++i) { s += gn_part[((long)b * tiles_per_b + i) * 2]; ss += gn_part[((long)b * tiles_per_b + i) * 2 + 1]; }
        const double n = (double)T * EF * C, mean = s / n, var = ss / n - mean * mean;
        stat[0] = (float)mean;
        stat[1] = (float)(1.0 / sqrt(var + (double)LN_EPS));
    }
    __syncthreads();
    const float mean = stat[0], rstd = stat[1];
    const int c4 = tid & 15;
    const float4 w4 = *reinterpret_cast<const float4*>(&gw[c4 * 4]);
    const float4 b4 = *reinterpret_cast<const float4*>(&gb[c4 * 4]);
    const long n4 = (long)T * EF * 16;
    float* zb = z + (long)b * T * EF * C;
    for (long i = (long)blockIdx.x * 256 + tid; i < n4; i += (long)gridDim.x * 256) {
        float4 v = *reinterpret_cast<float4*>(&zb[i * 4]);
        v.x = (v.x - mean) * rstd * w4.x + b4.x; v.y = (v.y - mean) * rstd * w4.y + b4.y;
        v.z = (v.z - mean) * rstd * w4.z + b4.z; v.w = (v.w - mean) * rstd * w4.w + b4.w;
        *reinterpret_cast<float4*>(&zb[i * 4]) = v;
        // block 0's intra axis call reads these rows channel-normalised and split (16 lanes = one row: i & 15 = tid & 15)
        if (xs_next) ln_split_store(v, xs_next, xs_next + rows_x * C, (long)b * T * EF * C + i * 4, true);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Axis paths of the espnet2 GridNetBlock (intra = along frequency, inter = along time):
//   LayerNorm over C -> unfold(4, stride 1) -> BiLSTM(256 -> 64) -> ConvTranspose1d(128 -> 64, 4) -> + residual.
// The 256-wide input projection is not on the recurrence, so it runs as one dense split-precision GEMM over all
// (sequence, window) rows (k_emb_gx: LN fused into the A staging, LN affine folded into the packed weights, output
// columns stored in the recurrent kernel's accumulator order); the recurrence (k_emb_lstm) then only carries the
// 64 x 256 hidden-to-hidden product, and the transposed conv is a K = 4*128 gather-GEMM with the residual add.
// ---------------------------------------------------------------------------------------------------------------
template <int RP>
__device__ __forceinline__ int e_slot(int blk, int row) { return (blk * RP + (row ^ (blk & 7))) * 8; }
template <int RP>
__device__ __forceinline__ int e_index(int row, int k) { return e_slot<RP>((k >> 5) * 4 + ((k >> 3) & 3), row) + (k & 7); }

template <int RP>
__device__ __forceinline__ void e_store_split4(_Float16* ahi, _Float16* alo, int row, int k0, float a, float b, float c, float d) {
    const float x[4] = {a, b, c, d};
    f16x4 h4, l4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        _Float16 h, l;
        split_hl(x[i], h, l);
        h4[i] = h;
        l4[i] = l;
    }
    const int idx = e_index<RP>(row, k0);
    *reinterpret_cast<f16x4*>(&ahi[idx]) = h4;
    *reinterpret_cast<f16x4*>(&alo[idx]) = l4;
}

template <int RP, int KS>
__device__ __forceinline__ f32x4 e_mma(const _Float16* ahi, const _Float16* alo, int m, int g4, int l15,
                                        const f16x8 (&wh)[KS], const f16x8 (&wl)[KS], float bias) {
    f32x4 am = f32x4{bias, bias, bias, bias}, ac = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int idx = e_slot<RP>(ks * 4 + g4, m * 16 + l15);
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
        const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[ks], am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[ks], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[ks], ac, 0, 0, 0);
    }
    // component-wise (a vector expression would lower to packed fp32: build.py)
    return f32x4{am[0] + ac[0] * (1.0f / ESPLIT), am[1] + ac[1] * (1.0f / ESPLIT), am[2] + ac[2] * (1.0f / ESPLIT),
                 am[3] + ac[3] * (1.0f / ESPLIT)};
}

// position row (in the [B][T][65][64] activation) of window element k of (sequence s, step p)
template <bool INTER>
__device__ __forceinline__ long pos_row(int s, int pk, int T) {
    if (INTER) return ((long)(s / EF) * T + pk) * EF + (s % EF);      // s = b*65 + f, pk = frame
    return (long)s * EF + pk;                                           // s = b*T + t,  pk = bin
}

// channel LayerNorm (no affine: folded into the packed input weights) + fp16 hi/lo split of every position row, once
// per axis call: xs = [hi image rows*64 halves | lo image rows*64 halves]; 16 lanes per 256-byte row
__global__ void __launch_bounds__(256) k_emb_lnsplit(const float* __restrict__ x, _Float16* __restrict__ xs, long rows) {
    const int tid = threadIdx.x, q = tid & 15;
    _Float16* xh = xs;
    _Float16* xl = xs + rows * C;
    for (long r = (long)blockIdx.x * 16 + (tid >> 4); r < rows; r += (long)gridDim.x * 16)
        ln_split_store(*reinterpret_cast<const float4*>(&x[r * C + q * 4]), xh, xl, r * C + q * 4, true);
}

#if defined(LH_LEGACY)   // round-1 axis path (k_emb_gx -> k_emb_lstm -> k_emb_convt_res behind lh_emb_axis): A/B lab + emulator
                         // builds only; the product library launches k_emb_rec / k_emb_convt2 (lh_emb_axis_fused)
// Gx[s*P + p][chunk*128 ..] = W_ih' [xhat(s, p) | xhat(s, p+1) | xhat(s, p+2) | xhat(s, p+3)] + b'
// One tile = 64 consecutive windows of ONE sequence: its 67 position rows are staged once (16-byte copies of the
// pre-split images, no conversion) and the unfold is just a row offset in the A-fragment address (k-step ks covers
// window element ks/2, channels 32*(ks&1)..+31).  grid (persistent, 4 column chunks of 128), weights in VGPRs.
constexpr int GX_RP = 81;                         // odd row pitch (16-byte slots): conflict-free 16-byte staging writes
constexpr int GX_ROWS = 64 + EKS - 1;             // 67 position rows per tile
constexpr int GX_NLD = (GX_ROWS * 8 + 255) / 256; // 3 x 16 bytes per thread and image
constexpr int GX_CSP = 132;
template <bool INTER>
__global__ void __launch_bounds__(256, 2) k_emb_gx(const _Float16* __restrict__ xs, const _Float16* __restrict__ w_pk,
                                                   const float* __restrict__ bias, float* __restrict__ gx, int nseq,
                                                   int P, int T, long rows_x) {
    constexpr int KS = 8;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[8 * GX_RP * 8];
    __shared__ __attribute__((aligned(16))) _Float16 alo[8 * GX_RP * 8];
    __shared__ __attribute__((aligned(16))) float cs[64 * GX_CSP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int chunk = blockIdx.y;
    const _Float16* xh = xs;
    const _Float16* xl = xs + rows_x * C;
    f16x8 wh[2][KS], wl[2][KS];
    float bz[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int nt = chunk * 8 + wave * 2 + i;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const _Float16* p = w_pk + ((long)(nt * KS + ks) * 64 + lane) * 16;
            wh[i][ks] = *reinterpret_cast<const f16x8*>(p);
            wl[i][ks] = *reinterpret_cast<const f16x8*>(p + 8);
        }
        bz[i] = bias[nt * 16 + l15];
    }
    const int L = P + EKS - 1;
    const int tps = (P + 63) / 64;                       // tiles per sequence
    const int ntiles = nseq * tps;
    f16x8 sh[GX_NLD], sl[GX_NLD];
    auto fetch = [&](int tile) {
        const int s = tile / tps, p0 = (tile % tps) * 64;
#pragma unroll
        for (int i = 0; i < GX_NLD; ++i) {
            const int e = min(tid + 256 * i, GX_ROWS * 8 - 1);
            const long off = pos_row<INTER>(s, min(p0 + (e >> 3), L - 1), T) * C + (e & 7) * 8;
            sh[i] = *reinterpret_cast<const f16x8*>(&xh[off]);
            sl[i] = *reinterpret_cast<const f16x8*>(&xl[off]);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile / tps, p0 = (tile % tps) * 64;
        const int valid = min(64, P - p0);
#pragma unroll
        for (int i = 0; i < GX_NLD; ++i) {
            const int e = tid + 256 * i;
            if (e < GX_ROWS * 8) {
                const int idx = ((e & 7) * GX_RP + (e >> 3)) * 8;
                *reinterpret_cast<f16x8*>(&ahi[idx]) = sh[i];
                *reinterpret_cast<f16x8*>(&alo[idx]) = sl[i];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
#pragma unroll 1
        for (int m = 0; m < 4; ++m) {
            f32x4 am[2], ac[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { am[i] = f32x4{bz[i], bz[i], bz[i], bz[i]}; ac[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int idx = (((ks & 1) * 4 + g4) * GX_RP + m * 16 + l15 + (ks >> 1)) * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    am[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[i][ks], am[i], 0, 0, 0);
                    ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[i][ks], ac[i], 0, 0, 0);
                    ac[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[i][ks], ac[i], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f32x4 acc = f32x4{am[i][0] + ac[i][0] * (1.0f / ESPLIT), am[i][1] + ac[i][1] * (1.0f / ESPLIT),
                                        am[i][2] + ac[i][2] * (1.0f / ESPLIT), am[i][3] + ac[i][3] * (1.0f / ESPLIT)};
#pragma unroll
                for (int r = 0; r < 4; ++r) cs[(m * 16 + g4 * 4 + r) * GX_CSP + (wave * 2 + i) * 16 + l15] = acc[r];
            }
        }
        __syncthreads();
        float* dst = gx + ((long)s * P + p0) * 512 + chunk * 128;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, rr = e >> 5, c4 = e & 31;
            if (rr < valid)
                *reinterpret_cast<float4*>(&dst[(long)rr * 512 + c4 * 4]) = *reinterpret_cast<const float4*>(&cs[rr * GX_CSP + c4 * 4]);
        }
    }
}

// recurrence: gates = Gx[row] + h W_hh^T ; grid (ceil(nseq/16), 2 directions), zero initial state
constexpr int EL_AP = 144;    // halves per h-image row (64 used): 288-byte stride keeps ds_read_b128 conflict-free
constexpr int EL_HP = 68;
__global__ void __launch_bounds__(256, 2) k_emb_lstm(const float* __restrict__ gx, const _Float16* __restrict__ w_pk,
                                                     float* __restrict__ h_out, int nseq, int P) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * 16 * EL_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * 16 * EL_AP];
    __shared__ __attribute__((aligned(16))) float hf[2 * 16 * EL_HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int dir = blockIdx.y, s0 = blockIdx.x * 16;
    const int unit = wave * 16 + l15;
    const int rl = tid >> 4, q = tid & 15;
    f16x8 wh[4][2], wl[4][2];
    {
        const _Float16* wp = w_pk + ((long)(dir * 4 + wave) * 8 * 64 + lane) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                wh[g][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(g * 2 + ks) * 64 * 16);
                wl[g][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(g * 2 + ks) * 64 * 16 + 8);
            }
    }
    auto step_pos = [&](int it) -> int { it = min(max(it, 0), P - 1); return dir ? (P - 1 - it) : it; };
    auto load_gx = [&](int it, float4 (&g)[4]) {           // this lane's (i,f,g,o) pre-activations of its 4 rows
        const int p = step_pos(it);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int s = min(s0 + g4 * 4 + r, nseq - 1);
            g[r] = *reinterpret_cast<const float4*>(&gx[((long)s * P + p) * 512 + dir * 256 + unit * 4]);
        }
    };
    for (int i = tid; i < 16 * EL_AP; i += 256) { ahi[i] = (_Float16)0.f; alo[i] = (_Float16)0.f; }
    for (int i = tid; i < 16 * EL_HP; i += 256) hf[i] = 0.f;
    float creg[4] = {0.f, 0.f, 0.f, 0.f};
    float4 gxr[4];
    load_gx(0, gxr);
    __syncthreads();
    constexpr float INV = 1.0f / ESPLIT;
    for (int it = 0; it < P; ++it) {
        const int cur = it & 1, nxt = cur ^ 1;
        {   // flush h_{it-1} (clamped: at it = 0 zeros go to the row of step 0 and are overwritten a step later)
            const int s = min(s0 + rl, nseq - 1);
            *reinterpret_cast<float4*>(&h_out[((long)s * P + step_pos(it - 1)) * 128 + dir * 64 + q * 4]) =
                *reinterpret_cast<const float4*>(&hf[(cur * 16 + rl) * EL_HP + q * 4]);
        }
        f32x4 am[4], ac[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            am[g] = f32x4{(&gxr[0].x)[g], (&gxr[1].x)[g], (&gxr[2].x)[g], (&gxr[3].x)[g]};
            ac[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        load_gx(it + 1, gxr);
        const int ro = (cur * 16 + l15) * EL_AP + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[ro + ks * 32]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[ro + ks * 32]);
#pragma unroll
            for (int g = 0; g < 4; ++g) am[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[g][ks], am[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) ac[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[g][ks], ac[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) ac[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[g][ks], ac[g], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float hv;
            lstm_cell(am[0][r] + ac[0][r] * INV, am[1][r] + ac[1][r] * INV, am[2][r] + ac[2][r] * INV,
                      am[3][r] + ac[3][r] * INV, creg[r], hv);
            const int row = g4 * 4 + r;
            _Float16 th, tl;
            split_hl(hv, th, tl);
            ahi[(nxt * 16 + row) * EL_AP + unit] = th;
            alo[(nxt * 16 + row) * EL_AP + unit] = tl;
            hf[(nxt * 16 + row) * EL_HP + unit] = hv;
        }
        __syncthreads();
    }
    if (s0 + rl < nseq)
        *reinterpret_cast<float4*>(&h_out[((long)(s0 + rl) * P + step_pos(P - 1)) * 128 + dir * 64 + q * 4]) =
            *reinterpret_cast<const float4*>(&hf[((P & 1) * 16 + rl) * EL_HP + q * 4]);
}

// out[r] = x[r] + b + sum_{k<4} Wt_k h[(s, q - k)]   (ConvTranspose1d(128 -> 64, 4, stride 1) + residual); 32-row tiles
template <bool INTER>
__global__ void __launch_bounds__(256, 2) k_emb_convt_res(const float* __restrict__ h, const _Float16* __restrict__ w_pk,
                                                          const float* __restrict__ bias, const float* __restrict__ x,
                                                          float* __restrict__ out, long rows, int P, int T) {
    constexpr int RP = 32, KS = 16, CSP = C + 4;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[KS * 4 * RP * 8];
    __shared__ __attribute__((aligned(16))) _Float16 alo[KS * 4 * RP * 8];
    __shared__ __attribute__((aligned(16))) float cs[RP * CSP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    f16x8 wh[KS], wl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const _Float16* p = w_pk + ((long)(wave * KS + ks) * 64 + lane) * 16;
        wh[ks] = *reinterpret_cast<const f16x8*>(p);
        wl[ks] = *reinterpret_cast<const f16x8*>(p + 8);
    }
    const float bz = bias[wave * 16 + l15];
    const int ntiles = (int)((rows + RP - 1) / RP);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = (long)tile * RP;
        __syncthreads();
        // A row = [h(q) | h(q-1) | h(q-2) | h(q-3)], 128 floats each (zeros outside 0..P-1): 32 x 4 x 32 float4
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * (half * 8 + i);
                const int rr = e >> 7, k = (e >> 5) & 3, c4 = e & 31;
                const long r = min(r0 + rr, rows - 1);
                int s, qq;
                if (INTER) { const long bt = r / EF; s = (int)(bt / T) * EF + (int)(r % EF); qq = (int)(bt % T); }
                else { s = (int)(r / EF); qq = (int)(r % EF); }
                const int p = qq - k;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p >= 0 && p < P) v[i] = *reinterpret_cast<const float4*>(&h[((long)s * P + p) * 128 + c4 * 4]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = tid + 256 * (half * 8 + i);
                const int rr = e >> 7, k = (e >> 5) & 3, c4 = e & 31;
                e_store_split4<RP>(ahi, alo, rr, k * 128 + c4 * 4, v[i].x, v[i].y, v[i].z, v[i].w);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < RP / 16; ++m) {
            const f32x4 acc = e_mma<RP, KS>(ahi, alo, m, g4, l15, wh, wl, bz);
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(m * 16 + g4 * 4 + r) * CSP + wave * 16 + l15] = acc[r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 256 * i, rr = e >> 4, c4 = e & 15;
            const long r = r0 + rr;
            if (r < rows) {
                const float4 cv = *reinterpret_cast<const float4*>(&cs[rr * CSP + c4 * 4]);
                const float4 xv = *reinterpret_cast<const float4*>(&x[r * C + c4 * 4]);
                *reinterpret_cast<float4*>(&out[r * C + c4 * 4]) = make_float4(cv.x + xv.x, cv.y + xv.y, cv.z + xv.z, cv.w + xv.w);
            }
        }
    }
}

#endif  // LH_LEGACY

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the axis path without the gate pre-activation round trip.  k_emb_gx wrote Gx = [rows x 512] fp32 (10.2 GB per
// axis pass at B = 64) and k_emb_lstm read it back: 20 of the 31 GB an axis pass moved.  k_emb_rec computes the input half
// INSIDE the recurrent kernel, one step ahead of the dependent chain (as k_intra_xp / k_inter_xp do for the separator):
//   * one workgroup (8 waves, one per CU: 256 registers each) = 16 sequences of ONE direction; the gate GEMM is
//     TRANSPOSED — weights are the MFMA A operand, rows ordered (unit, gate), the 16 sequences are the N axis — so a
//     lane's accumulator quad is (i, f, g, o) of one unit and the cell update is lane-local; wave w owns units
//     8w .. 8w+7 as two row tiles {8w + 2u + j}, so a lane's two cells are ADJACENT units and h leaves as one 4-byte
//     LDS store per half (hi pair, lo pair);
//   * [W_ih (4 window slots x 64 channels) | W_hh] of the wave's 32 rows live in VGPRs as f16x3 A fragments: 128 + 32
//     registers (the whole register file of the CU holds one direction's 327 KB of hi/lo weights — the reason for one
//     workgroup per CU);
//   * positions (the unfold's elements) stream through an 8-slot LDS ring of pre-split rows [hi 64 | lo 64] (k_emb_lnsplit's
//     images, 16-byte copies, two steps of global prefetch in registers); the window of step t+1 = ring slots t+1 .. t+4;
//   * processing order: the reverse direction runs the same code on mirrored positions (pi = L-1-p) with its window
//     slots packed in reverse tap order (embed_net.py pack_rec), so nothing in the loop depends on the direction;
//   * hidden states leave as fp16 hi | lo images [row][128] (forward units in columns 0..63, reverse 64..127) — the
//     operand format of the transposed-conv GEMM that follows (k_emb_convt2), which therefore stages 16-byte copies
//     instead of splitting every element four times.
// Gates are pre-scaled for v_exp_f32 (weights.py gate_prescale; lstm_cell_pre).
// ---------------------------------------------------------------------------------------------------------------
constexpr int ER_NT = 512;                 // 8 waves
constexpr int ER_RP = 144;                 // halves per LDS row: [hi 64 | lo 64 | pad 16] — 288-byte stride, conflict-free ds_read_b128
constexpr int ER_RING = 8;                 // position slots
constexpr int ER_POS = 16 * ER_RP;         // halves per position slot (16 sequences)
constexpr int ER_NFRAG = 2 * (EKS * 2 + 2) * 2;   // per wave: 2 tiles x (4 slots x 2 k-steps + 2 k-steps of W_hh) x (hi, lo) = 40

#if defined(ER_TRACE)            // timing probe build only (scripts/probe_emb_rec.py): s_memtime stamps of waves 0 and 4 of workgroup 5
__device__ unsigned long long er_trace_buf[2 * 64 * 8];
#define ER_STAMP(k) do { if (tr_on && it >= 8 && it < 72) tr[((it - 8) * 8) + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ER_STAMP(k) do { } while (0)
#endif

template <bool INTER>
__global__ void __launch_bounds__(ER_NT, 1) k_emb_rec(const _Float16* __restrict__ xs, const _Float16* __restrict__ w_pk,
                                                      const float* __restrict__ bias, _Float16* __restrict__ hs, int nseq,
                                                      int P, int T, long rows_x, int prio) {
    __shared__ __attribute__((aligned(16))) _Float16 ring[ER_RING * ER_POS];
    __shared__ __attribute__((aligned(16))) _Float16 himg[2 * ER_POS];
    __shared__ __attribute__((aligned(16))) float bsm[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int dir = blockIdx.y, s0 = blockIdx.x * 16;
    const int L = P + EKS - 1;
    const long hrows = (long)nseq * P;
#if defined(ER_TRACE)
    const bool tr_on = blockIdx.x == 5 && blockIdx.y == 0 && (tid & 255) == 0;
    unsigned long long* const tr = er_trace_buf + (tid >> 8) * 512;
#endif

    // resident A fragments
    f16x8 wxh[2][EKS][2], wxl[2][EKS][2], whh[2][2], whl[2][2];
    {
        const _Float16* wp = w_pk + (((long)(dir * 8 + wave) * ER_NFRAG) * 64 + lane) * 8;
        int f = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int k4 = 0; k4 < EKS; ++k4)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    wxh[j][k4][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(f++) * 64 * 8);
                    wxl[j][k4][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(f++) * 64 * 8);
                }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                whh[j][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(f++) * 64 * 8);
                whl[j][ks] = *reinterpret_cast<const f16x8*>(wp + (long)(f++) * 64 * 8);
            }
        }
    }
    // bias of this lane's rows (tile j: unit 8w + 2 g4 + j, gates 0..3 = accumulator rows 4 g4 + gate): kept in LDS and read
    // into the x-half accumulators at the top of every step — 8 registers the 256-register budget does not have
    if (tid < 256) bsm[tid] = bias[dir * 256 + tid];
    const int b_off = (wave * 8 + 2 * g4) * 4;
    auto bias_acc = [&](f32x4 (&acc)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float4 b4 = *reinterpret_cast<const float4*>(&bsm[b_off + j * 4]);
            acc[j] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
    };

    // row-wise roles.  Waves 0..3 stage positions: thread -> (sequence row, 16-byte piece: 0..7 of the hi image, 8..15 lo)
    const int rrow = (tid & 255) >> 4, piece = tid & 15;
    const int sq = min(s0 + rrow, nseq - 1);
    const bool loader = wave < 4;
    // position p of sequence sq sits at row pos_row(sq, 0) + p * (INTER ? 65 : 1): base pointer + stride, no index algebra
    // in the loop
    const _Float16* xsrc = xs + (piece < 8 ? 0 : rows_x * C) + (piece & 7) * 8 + pos_row<INTER>(sq, 0, T) * C;
    const int pstride = (INTER ? EF : 1) * C;
    auto load_pos = [&](int pi) -> f16x8 {                    // processing-order position pi (clamped), mirrored for dir 1
        pi = min(pi, L - 1);
        const int p = dir ? L - 1 - pi : pi;
        return *reinterpret_cast<const f16x8*>(&xsrc[(long)p * pstride]);
    };
    auto put_pos = [&](int pi, const f16x8& v) {
        *reinterpret_cast<f16x8*>(&ring[(pi & (ER_RING - 1)) * ER_POS + rrow * ER_RP + piece * 8]) = v;
    };
    // waves 4..7 write the hidden states of the previous step: himg row -> hi | lo images, this direction's 64 columns
    _Float16* hdst = hs + (piece < 8 ? 0 : hrows * 128) + dir * H + (piece & 7) * 8 + (long)(s0 + rrow) * P * 128;
    const bool hvalid = s0 + rrow < nseq;
    auto flush_h = [&](int it, int buf) {                     // h of processing step it
        if (hvalid) {
            const int t = dir ? P - 1 - it : it;
            *reinterpret_cast<f16x8*>(&hdst[(long)t * 128]) =
                *reinterpret_cast<const f16x8*>(&himg[buf * ER_POS + rrow * ER_RP + piece * 8]);
        }
    };

    // x half of the gates of processing step `it`: bias + sum over the 4 window slots (ring positions it .. it+3), as 8
    // groups (slot k4, k-step ks) of 6 MFMAs; x_frag(g) reads the group's B fragments
    const int frag_off = l15 * ER_RP + g4 * 8;
    auto x_frag = [&](int it, int g, f16x8& xh, f16x8& xl) {
        const _Float16* base = &ring[((it + (g >> 1)) & (ER_RING - 1)) * ER_POS + frag_off + (g & 1) * 32];
        xh = *reinterpret_cast<const f16x8*>(base);
        xl = *reinterpret_cast<const f16x8*>(base + 64);
    };
    auto gate_x = [&](int it, f32x4 (&am)[2], f32x4 (&ac)[2]) {             // plain form (prologue)
        ac[0] = ac[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        bias_acc(am);
        xp_for<8>([&](auto g_) {
            constexpr int g = decltype(g_)::value, k4 = g >> 1, ks = g & 1;
            f16x8 xh, xl;
            x_frag(it, g, xh, xl);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                am[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wxh[j][k4][ks], xh, am[j], 0, 0, 0);
                ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wxh[j][k4][ks], xl, ac[j], 0, 0, 0);
                ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wxl[j][k4][ks], xh, ac[j], 0, 0, 0);
            }
        });
    };

    // prologue: positions 0..5 into the ring, positions 6 and 7 in flight; h_{-1} = 0
    f16x8 stA = f16x8{0, 0, 0, 0, 0, 0, 0, 0}, stB = stA;
    if (loader) {
        f16x8 p0[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) p0[i] = load_pos(i);
        stA = load_pos(6);
        stB = load_pos(7);
#pragma unroll
        for (int i = 0; i < 6; ++i) put_pos(i, p0[i]);
    } else {
        *reinterpret_cast<f16x8*>(&himg[rrow * ER_RP + piece * 8]) = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float creg[2] = {0.f, 0.f};
    __syncthreads();
    // Two accumulator sets (hh products | cross terms), alternating by step parity: the x half of step it + 1 is accumulated into
    // set (it + 1) & 1 during step it, and step it + 1's MFMAs on the chain simply continue in the same registers — no sum, no copy
    // between the steps (16 vector instructions and 8 registers per step less than summing into a third set)
    f32x4 gm[2][2], gc[2][2];
    gate_x(0, gm[0], gc[0]);

    // One step, hand-ordered (hipcc's own schedule ran all 60 MFMAs, then the ~75 vector instructions of the two cell
    // updates with the matrix pipe idle, then the barrier: 3300 cycles per step against 2040 of MFMA issue for the two waves
    // of a SIMD).  Order: h fragments + first x group issued -> x group 0 (6 MFMAs, off the chain, while the h reads land)
    // -> the 12 MFMAs on the chain -> x groups 1..7 (42 MFMAs), each MFMA followed by one slice of the cell update and a
    // scheduling fence, the next group's fragments read one group ahead.
    auto step = [&](int it, f16x8& stg, auto par_) {
        constexpr int PAR = decltype(par_)::value;        // it & 1, compile time: selects the accumulator sets
        f32x4 (&am)[2] = gm[PAR], (&ac)[2] = gc[PAR], (&xm)[2] = gm[PAR ^ 1], (&xc)[2] = gc[PAR ^ 1];
        const int cur = it & 1;
        // (the last step still runs an x half — of a clamped window — whose result is dropped)
        ER_STAMP(0);
        f16x8 hh[2], hl[2], xh[2], xl[2];
        {
            const _Float16* base = &himg[cur * ER_POS + frag_off];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                hh[ks] = *reinterpret_cast<const f16x8*>(base + ks * 32);
                hl[ks] = *reinterpret_cast<const f16x8*>(base + 64 + ks * 32);
            }
        }
        x_frag(it + 1, 0, xh[0], xl[0]);
        xc[0] = xc[1] = f32x4{0.f, 0.f, 0.f, 0.f};        // x half of step it + 1 (am / ac: this step's, on the chain)
        bias_acc(xm);
        // cell update of the two units of this lane, cut into 29 slices (lstm_cell_pre's arithmetic, same operation order)
        float a[2][4], r[2][4], g2[2], cc[2], hv[2];
        _Float16 sh_[2], sl_[2];
        auto cell_op = [&](auto k_) {
            constexpr int k = decltype(k_)::value;
            if constexpr (k >= 0 && k < 26) {
                // the two cells' slices alternate: a slice's input is two slots old (each slice depends on the cell's previous
                // one — transcendental latency — so cell 0 then cell 1 made every slot wait for the one before)
                constexpr int j = k & 1, o = k >> 1;
                if constexpr (o == 0) { a[j][0] = am[j][0] + ac[j][0]; a[j][1] = am[j][1] + ac[j][1]; }
                if constexpr (o == 1) { a[j][2] = am[j][2] + ac[j][2]; a[j][3] = am[j][3] + ac[j][3]; }
                if constexpr (o == 2) { a[j][0] = __builtin_amdgcn_exp2f(a[j][0]); a[j][1] = __builtin_amdgcn_exp2f(a[j][1]); }
                if constexpr (o == 3) { a[j][2] = __builtin_amdgcn_exp2f(a[j][2]); a[j][3] = __builtin_amdgcn_exp2f(a[j][3]); }
                if constexpr (o == 4) { a[j][0] = 1.0f + a[j][0]; a[j][1] = 1.0f + a[j][1]; }
                if constexpr (o == 5) { a[j][2] = 1.0f + a[j][2]; a[j][3] = 1.0f + a[j][3]; }
                if constexpr (o == 6) { r[j][0] = __builtin_amdgcn_rcpf(a[j][0]); r[j][1] = __builtin_amdgcn_rcpf(a[j][1]); }
                if constexpr (o == 7) { r[j][2] = __builtin_amdgcn_rcpf(a[j][2]); r[j][3] = __builtin_amdgcn_rcpf(a[j][3]); }
                if constexpr (o == 8) { g2[j] = 2.0f * r[j][2] - 1.0f; }
                if constexpr (o == 9) { cc[j] = r[j][1] * creg[j] + r[j][0] * g2[j]; creg[j] = cc[j]; }
                if constexpr (o == 10) { cc[j] = __builtin_amdgcn_exp2f(-2.0f * LOG2E * cc[j]); }
                if constexpr (o == 11) { cc[j] = __builtin_amdgcn_rcpf(1.0f + cc[j]); }
                if constexpr (o == 12) { hv[j] = r[j][3] * (2.0f * cc[j] - 1.0f); }
            }
            if constexpr (k == 26) split_hl(hv[0], sh_[0], sl_[0]);
            if constexpr (k == 27) split_hl(hv[1], sh_[1], sl_[1]);
            if constexpr (k == 28) {
                typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                _Float16* hrow = &himg[(cur ^ 1) * ER_POS + l15 * ER_RP + wave * 8 + 2 * g4];
                *reinterpret_cast<f16x2*>(hrow) = f16x2{sh_[0], sh_[1]};
                *reinterpret_cast<f16x2*>(hrow + 64) = f16x2{sl_[0], sl_[1]};
            }
            if constexpr (k == 30) {                     // row-wise roles, inside the MFMA stream instead of behind it
                if (loader) {                            // position it + 6 (loaded two steps ago) -> ring; fetch it + 8
                    put_pos(it + 6, stg);
                    stg = load_pos(it + 8);
                } else if (it > 0) {
                    flush_h(it - 1, cur);
                }
            }
        };
        auto x_group = [&](auto g_, auto zip_) {          // 6 MFMAs of group g; with zip: cell slice + fence after each
            constexpr int g = decltype(g_)::value, k4 = g >> 1, ks = g & 1, b = g & 1;
            constexpr bool zip = decltype(zip_)::value;
            if constexpr (g + 1 < 8) x_frag(it + 1, g + 1, xh[b ^ 1], xl[b ^ 1]);
            xp_for<6>([&](auto i_) {
                constexpr int i = decltype(i_)::value, j = i / 3, pr = i % 3;
                if constexpr (pr == 0) xm[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wxh[j][k4][ks], xh[b], xm[j], 0, 0, 0);
                if constexpr (pr == 1) xc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wxh[j][k4][ks], xl[b], xc[j], 0, 0, 0);
                if constexpr (pr == 2) xc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wxl[j][k4][ks], xh[b], xc[j], 0, 0, 0);
                if constexpr (zip) {
                    constexpr int slot = (g - 1) * 6 + i;                  // 0 .. 41
                    cell_op(std::integral_constant<int, slot - 6>{});      // the chain's results are ~6 MFMAs old by then
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        };
        x_group(std::integral_constant<int, 0>{}, std::false_type{});
        __builtin_amdgcn_sched_barrier(0);
        ER_STAMP(1);
        // issue priority for the MFMAs on the chain (lh_set_tuning key 16): the SIMD's other wave is in its off-chain groups
        // half of the time, and the arbiter otherwise lets those in first (k_inter_xp: -22 % from the same switch)
        if (prio) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                am[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whh[j][ks], hh[ks], am[j], 0, 0, 0);
                ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whh[j][ks], hl[ks], ac[j], 0, 0, 0);
                ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whl[j][ks], hh[ks], ac[j], 0, 0, 0);
            }
        if (prio) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        ER_STAMP(2);
        xp_for<7>([&](auto g_) { x_group(std::integral_constant<int, decltype(g_)::value + 1>{}, std::true_type{}); });
        ER_STAMP(3);
        ER_STAMP(4);
        __syncthreads();
        ER_STAMP(5);
    };
    int it = 0;
    for (; it + 1 < P; it += 2) {
        step(it, stA, std::integral_constant<int, 0>{});
        step(it + 1, stB, std::integral_constant<int, 1>{});
    }
    if (it < P) step(it, stA, std::integral_constant<int, 0>{});
    if (!loader) flush_h(P - 1, P & 1);
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the INTER-axis recurrence for SMALL batches, one workgroup per (sequence, direction).  The tiled kernel above needs
// 16 sequences per workgroup: one enrollment clip is 65 x 2 / 16 = 9 workgroups walking 1248 dependent steps at ~1.4 us each —
// 1.75 ms per block, 5.2 of the 5.6 ms a single 5 s enrollment took, with 247 CUs dark.  Here every (sequence, direction) gets
// its own CU and the step is the quad-lane 256 x 64 mat-vec of the separator's batch-1 kernels (lh_quad.h: 0.39 us per step);
// the time axis is cut into chunks of 32 steps whose 256-wide input half (the unfold's four window slots x 64 channels, taps =
// row offsets into the staged position rows like k_emb_gx) runs as a split-precision MFMA GEMM in front of the chunk's
// recurrence; the chunk's hidden states leave as fp16 hi | lo rows (k_emb_convt2's operand format).  The reverse direction walks
// the natural positions downwards (weights in natural tap order: embed_net.py `_pack_axis`, not pack_rec's mirrored form).
//   xs      pre-split LayerNorm(x) images (k_emb_lnsplit / emit_split), rows_x rows of 64 halves, hi then lo
//   wih_pk  fp16 hi/lo B image [2 dirs x 16 ntiles][8 ksteps][64 lanes][16] of W_ih' (LN gamma folded), K = slot*64 + channel,
//           columns (direction, unit, gate);  bih [2][256] in the same column order (b_ih + b_hh + W_ih beta)
//   whh     fp32 [2][256][64], row 4 unit + gate
// ---------------------------------------------------------------------------------------------------------------
// Workgroups of FOUR waves (one per SIMD, every thread a recurrence thread), 52 KB of LDS and <= 128 registers: three workgroups per
// CU.  The step is latency, not issue — further sequences on the same CU cost little — and 3 x 256 resident workgroups take
// 2 B 65 <= 768 (B <= 5) in ONE round.  (First version: 512 threads, 64-step chunks, 200 registers = one per CU: B = 2's 260
// workgroups needed a second round for FOUR of them; second: two per CU: B = 4's 520 a second round for eight.)
constexpr int MV_TC = 32;                          // steps per chunk
constexpr int MV_ROWS = MV_TC + EKS - 1;           // 35 position rows per chunk
constexpr int MV_RP = 41;                          // odd row pitch of the staged rows (16-byte slots), like k_emb_gx
constexpr int MV_HP = H + 8;                       // fp16 row pitch of the chunk's hidden states (144 B: 16-byte aligned rows)

// MV_NT = 256 (three per CU) or 512 (two per CU; threads 256.. only work in the staging / GEMM phases and otherwise keep the step
// barrier company: 4 % faster while one workgroup per CU is all there is — a single enrollment)
template <int MV_NT>
__global__ void __launch_bounds__(MV_NT, MV_NT == IS_NR ? 3 : 2) k_emb_inter_mv(const _Float16* __restrict__ xs, const _Float16* __restrict__ wih_pk,
                                                        const float* __restrict__ bih, const float* __restrict__ whh,
                                                        _Float16* __restrict__ hs, int nseq, int P, int T, long rows_x) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[8 * MV_RP * 8];
    __shared__ __attribute__((aligned(16))) _Float16 alo[8 * MV_RP * 8];
    __shared__ __attribute__((aligned(16))) float gxs[MV_TC * IS_GP];      // input half of the gates, [step][column 4 u + g]
    __shared__ __attribute__((aligned(16))) _Float16 hh[MV_TC * MV_HP];    // h of the chunk's steps, hi | lo
    __shared__ __attribute__((aligned(16))) _Float16 hl[MV_TC * MV_HP];
    __shared__ __attribute__((aligned(16))) float hprev[2][H];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int seq = blockIdx.x >> 1, dir = blockIdx.x & 1;
    const int L = P + EKS - 1;
    const long hrows = (long)nseq * P;
    const _Float16* xh = xs;
    const _Float16* xl = xs + rows_x * C;

    const int unit = (tid & (IS_NR - 1)) >> 2, qs = tid & 3;             // recurrence role (threads < 256): hidden unit, k slice
    f32x2 wr[4][8];
    quad_load_w(whh + (long)dir * IS_GP * H, unit, qs, wr);
    const bool cell_lane = qs == 1 && tid < IS_NR;
    const float gscale = quad_gate_scale(qs);
    float c = 0.f;                                                         // QS_K2 x cell state; the embedder carries none
    if (tid < H) hprev[0][tid] = 0.f;
    int hb = 0;

    const int nchunk = (P + MV_TC - 1) / MV_TC;
    for (int ci = 0; ci < nchunk; ++ci) {
        // natural steps [p0, p0 + n) of this chunk; the reverse direction takes its chunks (and their steps) from the end
        const int p0 = dir ? max(P - (ci + 1) * MV_TC, 0) : ci * MV_TC;
        const int n = dir ? (P - ci * MV_TC) - p0 : min(MV_TC, P - p0);
        // ---- stage position rows p0 .. p0 + n + 2 (16-byte copies of the pre-split images)
        for (int e = tid; e < MV_ROWS * 8; e += MV_NT) {
            const long off = pos_row<true>(seq, min(p0 + (e >> 3), L - 1), T) * C + (e & 7) * 8;
            const int idx = ((e & 7) * MV_RP + (e >> 3)) * 8;
            *reinterpret_cast<f16x8*>(&ahi[idx]) = *reinterpret_cast<const f16x8*>(&xh[off]);
            *reinterpret_cast<f16x8*>(&alo[idx]) = *reinterpret_cast<const f16x8*>(&xl[off]);
        }
        __syncthreads();
        // ---- G_x[step][col] = b[col] + sum_{slot, c} xhat[step + slot][c] W'[col][slot*64 + c]: wave w owns column tiles 4w .. 4w+3;
        //      k-step outermost with the chunk's row tiles as accumulators: every weight fragment is fetched (L2) once per chunk and
        //      only one k-step of them is live (the 128-register budget of two workgroups per CU)
#pragma unroll 1
        for (int i = 0; i < 16 / (MV_NT / 64); ++i) {
            const int nt = (16 / (MV_NT / 64)) * wave + i;
            const float bz = bih[dir * IS_GP + nt * 16 + l15];
            f32x4 am[MV_TC / 16], ac[MV_TC / 16];
#pragma unroll
            for (int m = 0; m < MV_TC / 16; ++m) { am[m] = f32x4{bz, bz, bz, bz}; ac[m] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
            for (int ks = 0; ks < 8; ++ks) {
                const _Float16* p = wih_pk + ((long)((dir * 16 + nt) * 8 + ks) * 64 + lane) * 16;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(p);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(p + 8);
#pragma unroll
                for (int m = 0; m < MV_TC / 16; ++m) {
                    const int idx = (((ks & 1) * 4 + g4) * MV_RP + m * 16 + l15 + (ks >> 1)) * 8;
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
                    const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
                    am[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh, am[m], 0, 0, 0);
                    ac[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl, ac[m], 0, 0, 0);
                    ac[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh, ac[m], 0, 0, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < MV_TC / 16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) gxs[(m * 16 + g4 * 4 + r) * IS_GP + nt * 16 + l15] = am[m][r] + ac[m][r];
        }
        __syncthreads();
        // ---- recurrence over the chunk's steps (lh_quad.h); h_t also goes into the chunk's fp16 rows
        if (MV_NT > IS_NR && tid >= IS_NR) {
            for (int j = 0; j < n; ++j) QS_SYNC();
            hb ^= n & 1;
        } else
        for (int j = 0; j < n; ++j) {
            const int row = dir ? n - 1 - j : j;                             // natural step p0 + row
            const float gx = gscale * gxs[row * IS_GP + tid];
            const float hv = quad_step(wr, hprev[hb] + 16 * qs, gx, c, qs);
            if (cell_lane) {
                hprev[hb ^ 1][unit] = hv;
                _Float16 th, tl;
                split_hl(hv, th, tl);
                hh[row * MV_HP + unit] = th;
                hl[row * MV_HP + unit] = tl;
            }
            hb ^= 1;
            QS_SYNC();
        }
        // ---- the chunk's hidden states: hi | lo images [row][128], this direction's 64 columns, 16-byte pieces
        for (int e = tid; e < n * 16; e += MV_NT) {
            const int row = e >> 4, piece = e & 15;
            _Float16* dst = hs + (piece < 8 ? 0 : hrows * 128) + ((long)seq * P + p0 + row) * 128 + dir * H + (piece & 7) * 8;
            *reinterpret_cast<f16x8*>(dst) = *reinterpret_cast<const f16x8*>(&(piece < 8 ? hh : hl)[row * MV_HP + (piece & 7) * 8]);
        }
        __syncthreads();                                                     // hh / hl / gxs / the staged rows are rewritten
    }
}

// out[r] = x[r] + b + sum_{k<4} Wt_k h[(s, q - k)]: ConvTranspose1d(128 -> 64, 4, stride 1) + residual from k_emb_rec's
// fp16 hi | lo hidden-state images.  One tile = 64 consecutive output positions of ONE sequence: its 67 rows of h are
// staged once (16-byte copies, no conversion; zero rows outside 0..P-1) and the transposed conv's taps are row offsets
// in the A-fragment address (k-step ks covers tap ks / 4, hidden columns 32 (ks & 3) ..), like k_emb_gx's unfold.
#if defined(CT_ODD_PITCH)
__device__ __forceinline__ constexpr int ct_swz(int) { return 0; }
#else
__device__ __forceinline__ constexpr int ct_swz(int kb) { return (kb >> 1) & 3; }     // (see CtShape::RP)
#endif
template <bool INTER>
struct CtShape {                                   // 16-row MFMA tiles per workgroup tile: the intra axis has 65 positions per
    static constexpr int MT = INTER ? 4 : 5;       // sequence — one tile of 80 (5 row tiles) instead of 64 + 1 (8 row tiles)
    static constexpr int RT = 16 * MT;             // output positions per tile
    static constexpr int ROWS = RT + EKS - 1;      // staged h rows
#if defined(CT_ODD_PITCH)
    static constexpr int RP = ROWS | 1;            // odd row pitch in 16-byte slots
#else
    // row pitch in 16-byte slots: a multiple of 16, row r of k-block plane kb in slot kb * RP + (r ^ ct_swz(kb)).  A fragment
    // read's ds_read_b128 lane group is rows {0-3, 12-15} + r0 of plane kb and rows {4-11} + r0 of plane kb + 1 (kb even):
    // complementary residues mod 16 whatever the tap offset r0 is, PROVIDED both planes sit at the same residue and share
    // their XOR — the odd pitch of round 4 (conflict-free staging stores) made every fragment read 2-way conflicted
    // (SQ_LDS_BANK_CONFLICT 0.46 of the LDS cycles, the LDS 0.5 busy: with four waves re-reading the same A fragments the
    // MFMA phase needed more LDS cycles than matrix cycles).  The staging stores (eight planes of one row per 8-lane group)
    // are 2-way now: 16 array cycles against the 13 a ds_write_b128 costs anyway.
    static constexpr int RP = (ROWS + 15) / 16 * 16;
#endif
    static constexpr int NXP = INTER ? MT : 3;     // residual rows prefetched across the MFMA phase (the 80-row tile has registers for 3 of 5)
    static constexpr int NLD = (ROWS * 16 + 255) / 256;
};
template <bool INTER>
__global__ void __launch_bounds__(256, 2) k_emb_convt2(const _Float16* __restrict__ hs, const _Float16* __restrict__ w_pk,
                                                       const float* __restrict__ bias, const float* __restrict__ x,
                                                       float* __restrict__ out, _Float16* __restrict__ xs_next, long rows_x,
                                                       int nseq, int P, int T) {
    // xs_next != NULL: the rows also leave channel-normalised and split (k_emb_lnsplit's images) for the NEXT axis pass, whose
    // own normalisation launch (a read + a write of the whole activation) then drops out
    using S = CtShape<INTER>;
    constexpr int KS = 16, CSP = C + 4, MT = S::MT, RT = S::RT, ROWS = S::ROWS, RP = S::RP, NLD = S::NLD, NXP = S::NXP;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[16 * RP * 8];
    __shared__ __attribute__((aligned(16))) _Float16 alo[16 * RP * 8];
    __shared__ __attribute__((aligned(16))) float cs[RT * CSP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    f16x8 wh[KS], wl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const _Float16* p = w_pk + ((long)(wave * KS + ks) * 64 + lane) * 16;
        wh[ks] = *reinterpret_cast<const f16x8*>(p);
        wl[ks] = *reinterpret_cast<const f16x8*>(p + 8);
    }
    const float bz = bias[wave * 16 + l15];
    const int L = P + EKS - 1;
    const int tps = (L + RT - 1) / RT;
    const int ntiles = nseq * tps;
    const long hrows = (long)nseq * P;
    const _Float16* hh = hs;
    const _Float16* hl = hs + hrows * 128;
    f16x8 sh[NLD], sl[NLD];
    auto fetch = [&](int tile) {
        const int s = tile / tps, q0 = (tile % tps) * RT;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = min(tid + 256 * i, ROWS * 16 - 1);
            const int p = q0 - (EKS - 1) + (e >> 4);
            sh[i] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            sl[i] = sh[i];
            if (p >= 0 && p < P) {
                const long off = ((long)s * P + p) * 128 + (e & 15) * 8;
                sh[i] = *reinterpret_cast<const f16x8*>(&hh[off]);
                sl[i] = *reinterpret_cast<const f16x8*>(&hl[off]);
            }
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int s = tile / tps, q0 = (tile % tps) * RT;
        const int valid = min(RT, L - q0);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i;
            if (e < ROWS * 16) {
                const int idx = ((e & 15) * RP + ((e >> 4) ^ ct_swz(e & 15))) * 8;
                *reinterpret_cast<f16x8*>(&ahi[idx]) = sh[i];
                *reinterpret_cast<f16x8*>(&alo[idx]) = sl[i];
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
        // the residual rows of this tile go out HERE and are in flight across the MFMA phase (loaded in the epilogue their
        // latency was exposed once per tile: -13 % / -6 % of the intra / inter launch, profiles/r05k_convt2_ab.txt)
        float4 xpre[NXP];
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int rr = (tid + 256 * i) >> 4;
            xpre[i] = *reinterpret_cast<const float4*>(&x[pos_row<INTER>(s, q0 + (rr < valid ? rr : 0), T) * C + (tid & 15) * 4]);
        }
#pragma unroll 1
        for (int m = 0; m < MT; ++m) {
            if (m * 16 >= valid) break;                  // (workgroup-uniform) row tiles past the sequence's last position
            f32x4 am = f32x4{bz, bz, bz, bz}, ac = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // output row m*16 + l15 = position q0 + that; tap k = ks >> 2 reads h row q - k = staged row (.. + 3 - k)
                const int idx = (((ks & 3) * 4 + g4) * RP + ((m * 16 + l15 + (EKS - 1) - (ks >> 2)) ^ ct_swz((ks & 3) * 4 + g4))) * 8;
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[ks], am, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[ks], ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[ks], ac, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(m * 16 + g4 * 4 + r) * CSP + wave * 16 + l15] = am[r] + ac[r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int e = tid + 256 * i, rr = e >> 4, c4 = e & 15;
            const bool ok = rr < valid;                  // (uniform over the 16 lanes of a row)
            const long r = pos_row<INTER>(s, q0 + (ok ? rr : 0), T);
            const float4 cv = *reinterpret_cast<const float4*>(&cs[rr * CSP + c4 * 4]);
            const float4 xv = i < NXP ? xpre[i < NXP ? i : 0] : *reinterpret_cast<const float4*>(&x[r * C + c4 * 4]);
            const float4 ov = make_float4(cv.x + xv.x, cv.y + xv.y, cv.z + xv.z, cv.w + xv.w);
            if (ok) *reinterpret_cast<float4*>(&out[r * C + c4 * 4]) = ov;
            if (xs_next) ln_split_store(ov, xs_next, xs_next + rows_x * C, r * C + c4 * 4, ok);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention branch of the espnet2 GridNetBlock.  Frame kernels (one [65 x 64] frame per iteration, persistent):
//   k_emb_qkv   per head h: Conv2d 1x1 (64 -> 8 | 8 | 16) + PReLU (own slope) + LayerNorm over (c, f) with [c][f]
//               affine; written head-major [4B][T][d*65] with flat index f*d + c (the order is internal: Q.K is
//               order-invariant and the V order is undone by the fused head merge of k_emb_attn)
//   k_emb_proj  Conv2d 1x1 (64 -> 64) + PReLU + LayerNorm over (c, f) + residual
// ---------------------------------------------------------------------------------------------------------------
constexpr int EFR_RP = 80;                       // 5 row tiles cover 65 bins
constexpr int EFR_A = 2 * 4 * EFR_RP * 8;
constexpr int EFR_NLD = (EF * 16 + 255) / 256;   // 5 float4 per thread
constexpr int ENQKV = NH * (2 * EE + VD);        // 128 output columns: Q (h*8+e) | K | V (h*16+v)

__device__ __forceinline__ void efr_load(const float* __restrict__ src, int tid, float4 (&stg)[EFR_NLD]) {
#pragma unroll
    for (int i = 0; i < EFR_NLD; ++i) {
        const int e = min(tid + 256 * i, EF * 16 - 1);
        stg[i] = *reinterpret_cast<const float4*>(&src[(e >> 4) * C + (e & 15) * 4]);
    }
}
__device__ __forceinline__ void efr_store(_Float16* ahi, _Float16* alo, int tid, const float4 (&stg)[EFR_NLD]) {
#pragma unroll
    for (int i = 0; i < EFR_NLD; ++i) {
        const int e = tid + 256 * i;
        if (e < EF * 16) e_store_split4<EFR_RP>(ahi, alo, e >> 4, (e & 15) * 4, stg[i].x, stg[i].y, stg[i].z, stg[i].w);
    }
}
__device__ __forceinline__ void efr_zero_pad(_Float16* ahi, _Float16* alo, int tid) {
    for (int e = tid; e < (EFR_RP - EF) * 16; e += 256) e_store_split4<EFR_RP>(ahi, alo, EF + (e >> 4), (e & 15) * 4, 0.f, 0.f, 0.f, 0.f);
}

// one wave: LayerNorm over ys[f][col0 + c], f < 65, c < D (flat index i = f*D + c), affine gw/gb indexed by i
template <int D>
__device__ __forceinline__ void e_ln_head(const float* ys, int yp, int col0, const float* __restrict__ gw,
                                          const float* __restrict__ gb, float* __restrict__ dst, int lane) {
    constexpr int N = EF * D, IT = (N + 63) / 64;
    auto at = [&](int k) -> float { const int i = lane + 64 * k; return i < N ? ys[(i / D) * yp + col0 + (i % D)] : 0.f; };
    // the affine of this lane's slots goes into registers FIRST, in flight under the two reductions: read inside the store
    // loop, hipcc ordered every (gw[i], gb[i]) pair behind the previous dst[i] store (it does not carry the kernel's
    // __restrict__ through the inlined lambda): 17 rounds of load -> s_waitcnt vmcnt(0) -> store per frame and head, the
    // exposed L2 latency of which was most of k_emb_qkv's frame time (round 5: 1.49 -> see profiles/r05k)
    float aw[IT], ab[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = min(lane + 64 * k, N - 1);
        aw[k] = gw[i];
        ab[k] = gb[i];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) s += at(k);
    const float mean = wave_sum(s) * (1.0f / N);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) { const float dv = at(k) - mean; if (lane + 64 * k < N) v += dv * dv; }
    const float rstd = rsqrtf(wave_sum(v) * (1.0f / N) + LN_EPS);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = lane + 64 * k;
        if (i < N) dst[i] = (at(k) - mean) * rstd * aw[k] + ab[k];
    }
}

// same LayerNorm, output as fp16 hi/lo rows of EQP halves (zero-padded past N) — the attention GEMM operand format
constexpr int EQP = 544;                         // 520 features padded to 17 k-steps of 32
template <int D>
__device__ __forceinline__ void e_ln_head_split(const float* ys, int yp, int col0, const float* __restrict__ gw,
                                                const float* __restrict__ gb, _Float16* __restrict__ dh,
                                                _Float16* __restrict__ dl, int lane) {
    constexpr int N = EF * D, IT = (EQP + 63) / 64;
    static_assert(N <= EQP, "row pitch");
    auto at = [&](int k) -> float { const int i = lane + 64 * k; return i < N ? ys[(i / D) * yp + col0 + (i % D)] : 0.f; };
    float aw[IT], ab[IT];                          // affine first, in flight under the reductions (see e_ln_head)
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = min(lane + 64 * k, N - 1);
        aw[k] = gw[i];
        ab[k] = gb[i];
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) s += at(k);
    const float mean = wave_sum(s) * (1.0f / N);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) { const float dv = at(k) - mean; if (lane + 64 * k < N) v += dv * dv; }
    const float rstd = rsqrtf(wave_sum(v) * (1.0f / N) + LN_EPS);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = lane + 64 * k;
        if (i < EQP) {
            const float o = i < N ? (at(k) - mean) * rstd * aw[k] + ab[k] : 0.f;
            _Float16 h, l;
            split_hl(o, h, l);
            dh[i] = h;
            dl[i] = l;
        }
    }
}

__global__ void __launch_bounds__(256, 2) k_emb_qkv(const float* __restrict__ y, const _Float16* __restrict__ w_pk,
                                                    const float* __restrict__ bias, const float* __restrict__ slopes,
                                                    const float* __restrict__ lnq_w, const float* __restrict__ lnq_b,
                                                    const float* __restrict__ lnk_w, const float* __restrict__ lnk_b,
                                                    const float* __restrict__ lnv_w, const float* __restrict__ lnv_b,
                                                    _Float16* __restrict__ q, _Float16* __restrict__ k, float* __restrict__ v,
                                                    int B, int T) {
    constexpr int YP = ENQKV + 1;
    const long img = (long)NH * B * T * EQP;              // q, k: [hi image | lo image], each [4B][T][544]
    __shared__ __attribute__((aligned(16))) _Float16 ahi[EFR_A];
    __shared__ __attribute__((aligned(16))) _Float16 alo[EFR_A];
    __shared__ float ys[EF * YP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    f16x8 wh0[2], wl0[2], wh1[2], wl1[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const _Float16* p0 = w_pk + ((long)(wave * 2 + ks) * 64 + lane) * 16;
        const _Float16* p1 = w_pk + ((long)((wave + 4) * 2 + ks) * 64 + lane) * 16;
        wh0[ks] = *reinterpret_cast<const f16x8*>(p0); wl0[ks] = *reinterpret_cast<const f16x8*>(p0 + 8);
        wh1[ks] = *reinterpret_cast<const f16x8*>(p1); wl1[ks] = *reinterpret_cast<const f16x8*>(p1 + 8);
    }
    const int c0 = wave * 16 + l15, c1 = c0 + 64;
    const float bz0 = bias[c0], bz1 = bias[c1], a0 = slopes[c0], a1 = slopes[c1];
    efr_zero_pad(ahi, alo, tid);
    const int nframes = B * T;
    float4 stg[EFR_NLD];
    if ((int)blockIdx.x < nframes) efr_load(y + (long)blockIdx.x * EF * C, tid, stg);
    for (int fr = blockIdx.x; fr < nframes; fr += gridDim.x) {
        const int b = fr / T, t = fr % T;
        efr_store(ahi, alo, tid, stg);
        __syncthreads();
        if (fr + (int)gridDim.x < nframes) efr_load(y + (long)(fr + gridDim.x) * EF * C, tid, stg);
#pragma unroll 1
        for (int m = 0; m < EFR_RP / 16; ++m) {
            const f32x4 r0 = e_mma<EFR_RP, 2>(ahi, alo, m, g4, l15, wh0, wl0, bz0);
            const f32x4 r1 = e_mma<EFR_RP, 2>(ahi, alo, m, g4, l15, wh1, wl1, bz1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + g4 * 4 + r;
                if (row < EF) { ys[row * YP + c0] = prelu_f(r0[r], a0); ys[row * YP + c1] = prelu_f(r1[r], a1); }
            }
        }
        __syncthreads();
        const int hd = wave, ln = lane + (fr >> 30);          // (fr >> 30) = 0: blocks LICM of the slot addresses
        const long row = ((long)hd * B + b) * T + t;           // head-major batch order [nh*B] like espnet2's torch.cat
        e_ln_head_split<EE>(ys, YP, hd * EE, lnq_w + hd * EDQK, lnq_b + hd * EDQK, q + row * EQP, q + img + row * EQP, ln);
        e_ln_head_split<EE>(ys, YP, NH * EE + hd * EE, lnk_w + hd * EDQK, lnk_b + hd * EDQK, k + row * EQP, k + img + row * EQP, ln);
        e_ln_head<VD>(ys, YP, 2 * NH * EE + hd * VD, lnv_w + hd * EDV, lnv_b + hd * EDV, v + row * EDV, ln);
    }
}

__global__ void __launch_bounds__(256, 2) k_emb_proj(const float* __restrict__ merged, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ bias, const float* __restrict__ slope,
                                                     const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                     const float* __restrict__ y2, float* __restrict__ out,
                                                     _Float16* __restrict__ xs_next, int nframes) {
    constexpr int YP = C + 4, N = EF * C, N4 = N / 4, NSLOT = (N4 + 255) / 256;     // 1040 float4 -> 5 slots
    __shared__ __attribute__((aligned(16))) _Float16 ahi[EFR_A];
    __shared__ __attribute__((aligned(16))) _Float16 alo[EFR_A];
    __shared__ __attribute__((aligned(16))) float ys[EF * YP];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    f16x8 wh[2], wl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const _Float16* p = w_pk + ((long)(wave * 2 + ks) * 64 + lane) * 16;
        wh[ks] = *reinterpret_cast<const f16x8*>(p); wl[ks] = *reinterpret_cast<const f16x8*>(p + 8);
    }
    const float bz = bias[wave * 16 + l15], a = slope[0];
    float4 pw[NSLOT], pb[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int i = min(tid + 256 * k, N4 - 1);
        pw[k] = *reinterpret_cast<const float4*>(&lnw[i * 4]);
        pb[k] = *reinterpret_cast<const float4*>(&lnb[i * 4]);
    }
    efr_zero_pad(ahi, alo, tid);
    float4 stg[EFR_NLD];
    if ((int)blockIdx.x < nframes) efr_load(merged + (long)blockIdx.x * N, tid, stg);
    for (int fidx = blockIdx.x; fidx < nframes; fidx += gridDim.x) {
        const long fr = (long)fidx * N;
        efr_store(ahi, alo, tid, stg);
        __syncthreads();
        if (fidx + (int)gridDim.x < nframes) efr_load(merged + (long)(fidx + gridDim.x) * N, tid, stg);
        float4 rv[NSLOT];
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) rv[k] = *reinterpret_cast<const float4*>(&y2[fr + (long)min(tid + 256 * k, N4 - 1) * 4]);
#pragma unroll 1
        for (int m = 0; m < EFR_RP / 16; ++m) {
            const f32x4 acc = e_mma<EFR_RP, 2>(ahi, alo, m, g4, l15, wh, wl, bz);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + g4 * 4 + r;
                if (row < EF) ys[row * YP + wave * 16 + l15] = prelu_f(acc[r], a);
            }
        }
        __syncthreads();
        float4 vv[NSLOT];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const int i = min(tid + 256 * k, N4 - 1);
            vv[k] = *reinterpret_cast<const float4*>(&ys[(i >> 4) * YP + (i & 15) * 4]);
            if (tid + 256 * k < N4) s += vv[k].x + vv[k].y + vv[k].z + vv[k].w;
        }
        const float mean = block_sum_256(s, red) * (1.0f / N);
        float vs = 0.f;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const float dx = vv[k].x - mean, dy = vv[k].y - mean, dz = vv[k].z - mean, dw = vv[k].w - mean;
            if (tid + 256 * k < N4) vs += dx * dx + dy * dy + dz * dz + dw * dw;
        }
        const float rstd = rsqrtf(block_sum_256(vs, red) * (1.0f / N) + LN_EPS);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const int i = tid + 256 * k;
            const bool ok = i < N4;                      // (uniform over the 16 lanes of a bin's row: N4 = 65 * 16)
            float4 o;
            o.x = rv[k].x + (vv[k].x - mean) * rstd * pw[k].x + pb[k].x;
            o.y = rv[k].y + (vv[k].y - mean) * rstd * pw[k].y + pb[k].y;
            o.z = rv[k].z + (vv[k].z - mean) * rstd * pw[k].z + pb[k].z;
            o.w = rv[k].w + (vv[k].w - mean) * rstd * pw[k].w + pb[k].w;
            if (ok) *reinterpret_cast<float4*>(&out[fr + i * 4]) = o;
            // the next block's intra pass reads these rows channel-normalised and split: emit that form here too
            if (xs_next) ln_split_store(o, xs_next, xs_next + (long)nframes * N, fr + (long)min(i, N4 - 1) * 4, ok);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Full attention over all T frames (no mask) as two batched split-precision GEMMs with the score matrix materialised
// (T x T fp32 per (head, utterance) = 6.3 MB; with V rows of 1040 features a flash-style kernel cannot hold a
// useful query tile of O in registers, and streaming K/V per 16 queries is L2-bound):
//   k_emb_vt        V [T][1040] fp32 -> V^T hi/lo [1040][Tp] fp16 (keys contiguous = MFMA k axis), zero-padded keys
//   k_gemm_nt<0>    S = Q K^T / sqrt(520)         A = Q hi/lo [T][544], B = K hi/lo [T][544]
//   k_emb_softmax   P = softmax(S) rows as fp16 hi/lo [T][Tp], zero-padded keys
//   k_gemm_nt<1>    O = P V, stored with the head merge fused: merged[b][t][f][h*16 + v] = O[h*B + b][t][f*16 + v]
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_emb_vt(const float* __restrict__ v, _Float16* __restrict__ vt, int T, int Tp,
                                                long img) {
    __shared__ float tl[64][65];
    const int tid = threadIdx.x;
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64, bh = blockIdx.z;
    const float* vb = v + (long)bh * T * EDV;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i, tt = e >> 6, cc = e & 63;
        tl[tt][cc] = (t0 + tt < T && c0 + cc < EDV) ? vb[(long)(t0 + tt) * EDV + c0 + cc] : 0.f;
    }
    __syncthreads();
    _Float16* oh = vt + (long)bh * EDV * Tp;
    _Float16* ol = oh + img;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, cc = e >> 4, t4 = (e & 15) * 4;
        if (c0 + cc < EDV) {
            f16x4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = tl[t4 + j][cc];
                _Float16 h, l;
                split_hl(x, h, l);
                h4[j] = h;
                l4[j] = l;
            }
            const long o = (long)(c0 + cc) * Tp + t0 + t4;
            *reinterpret_cast<f16x4*>(&oh[o]) = h4;
            *reinterpret_cast<f16x4*>(&ol[o]) = l4;
        }
    }
}

// C[M x N] = A[M x K] B[N x K]^T per batch, operands as fp16 hi/lo images (lo image at +imgA / +imgB halves), K a
// multiple of 32.  128 x 128 tile per workgroup, 4 waves as 2 x 2 (each 64 x 64 = 4 x 4 MFMA tiles, two fp32
// accumulator sets), one 32-wide k-step per stage: global -> registers -> double-buffered LDS (one barrier per
// stage), A/B fragments by conflict-free ds_read_b128.  Workgroups of one batch stay on one XCD (its Q/K or P/V^T
// panel is then read from HBM once per L2 instead of once per XCD).
constexpr int GM_RP = 130;                        // rows per 16-byte k-block plane (+2: staging writes conflict-free)
constexpr int GM_IMG = 4 * GM_RP * 8;             // halves per (operand, hi|lo) stage image
#if defined(LH_LEGACY)   // the GEMM as it shipped in rounds 1-4 (an exec-masked branch around every MFMA): A/B lab + emulator builds
template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_nt(const _Float16* __restrict__ A, const _Float16* __restrict__ Bm,
                                                    float* __restrict__ Cm, int M, int N, int K, int lda, int ldb,
                                                    long strideA, long strideB, long imgA, long imgB, int ldc,
                                                    long strideC, float scale, int nbatch, int tiles_m, int tiles_n,
                                                    int Bn, int T) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[2][4][GM_IMG];      // [stage][A hi, A lo, B hi, B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware decode: consecutive workgroup ids go round-robin over the 8 XCDs; keep a batch on one XCD
    const int ntile = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int batch = (j / ntile) * 8 + xcd, tile = j % ntile;
    if (batch >= nbatch) return;
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    const _Float16* Ab = A + (long)batch * strideA;
    const _Float16* Bb = Bm + (long)batch * strideB;

    // staging: 128 rows x 4 k-blocks of 16 bytes per image = 512 items, 2 per thread
    const int sr = tid >> 2, sb = tid & 3;
    const long a_off0 = (long)min(m0 + sr, M - 1) * lda + sb * 8, a_off1 = (long)min(m0 + sr + 64, M - 1) * lda + sb * 8;
    const long b_off0 = (long)min(n0 + sr, N - 1) * ldb + sb * 8, b_off1 = (long)min(n0 + sr + 64, N - 1) * ldb + sb * 8;
    const int s_idx0 = (sb * GM_RP + sr) * 8, s_idx1 = (sb * GM_RP + sr + 64) * 8;
    f16x8 st[8];
    auto fetch = [&](int k0) {
        st[0] = *reinterpret_cast<const f16x8*>(Ab + a_off0 + k0);        st[1] = *reinterpret_cast<const f16x8*>(Ab + a_off1 + k0);
        st[2] = *reinterpret_cast<const f16x8*>(Ab + imgA + a_off0 + k0); st[3] = *reinterpret_cast<const f16x8*>(Ab + imgA + a_off1 + k0);
        st[4] = *reinterpret_cast<const f16x8*>(Bb + b_off0 + k0);        st[5] = *reinterpret_cast<const f16x8*>(Bb + b_off1 + k0);
        st[6] = *reinterpret_cast<const f16x8*>(Bb + imgB + b_off0 + k0); st[7] = *reinterpret_cast<const f16x8*>(Bb + imgB + b_off1 + k0);
    };
    f32x4 am[4][4], ac[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) { am[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; ac[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    fetch(0);
    const int nk = K / 32;
    const int jn_live = min(4, max(0, (N - (n0 + wn * 64) + 15) / 16));      // 16-column sub-tiles of this wave with any column < N
    // (round 5, reading the ISA: derived from threadIdx this wave-uniform value is treated as divergent and EVERY MFMA of the
    // loop sits in its own exec-masked branch — s_and_saveexec + s_cbranch around each of the 48; k_gemm_nt2 below fixes it.
    // This kernel is kept as it shipped in rounds 1-4 for the A/B, lh_set_tuning(17, 0).)
#pragma unroll 1
    for (int ks = 0; ks < nk; ++ks) {
        _Float16* buf = &sm[ks & 1][0][0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f16x8*>(&buf[i * GM_IMG + s_idx0]) = st[2 * i];
            *reinterpret_cast<f16x8*>(&buf[i * GM_IMG + s_idx1]) = st[2 * i + 1];
        }
        __syncthreads();
        if (ks + 1 < nk) fetch((ks + 1) * 32);
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int idx = (g4 * GM_RP + wn * 64 + jn * 16 + l15) * 8;
            bh[jn] = *reinterpret_cast<const f16x8*>(&buf[2 * GM_IMG + idx]);
            bl[jn] = *reinterpret_cast<const f16x8*>(&buf[3 * GM_IMG + idx]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = (g4 * GM_RP + wm * 64 + i * 16 + l15) * 8;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&buf[idx]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&buf[GM_IMG + idx]);
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                if (jn < jn_live) {      // (wave-uniform) N = 1040 leaves 112 of the last tile's 128 columns empty: no MFMAs for those
                    am[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[jn], am[i][jn], 0, 0, 0);
                    ac[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[jn], ac[i][jn], 0, 0, 0);
                    ac[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[jn], ac[i][jn], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: lane holds rows g4*4 + r, column l15 of each 16 x 16 tile -> 64-byte row segments
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int col = n0 + wn * 64 + jn * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + g4 * 4 + r;
                if (row < M && col < N) {
                    const float val = (am[i][jn][r] + ac[i][jn][r] * (1.0f / ESPLIT)) * scale;
                    if (EPI == 0) {
                        Cm[(long)batch * strideC + (long)row * ldc + col] = val;
                    } else {                     // batch = h*Bn + b, row = frame, col = f*16 + v
                        const int hd = batch / Bn, b = batch % Bn;
                        Cm[(((long)b * T + row) * EF + (col >> 4)) * C + hd * VD + (col & 15)] = val;
                    }
                }
            }
        }
}

#endif  // LH_LEGACY

// Round 5: k_gemm_nt3 = the same tiling with two defects removed (read off the ISA).
//  (1) `jn_live` (how many 16-column sub-tiles of the wave hold a live column) is wave-uniform but derived from threadIdx, so
//      the compiler took it for divergent and put EVERY MFMA of the loop into its own exec-masked branch (s_and_saveexec +
//      s_cbranch around each of the 48; 137 s_and_saveexec in the kernel).  Now: RAGGED = false computes every sub-tile
//      (straight-line MFMAs; columns >= N run on a clamped duplicate row and are not stored), RAGGED = true skips dead
//      sub-tiles with SCALAR branches (`jn_live` through v_readfirstlane).  Two kernels, not two paths in one: the register
//      allocator sees one loop each.  The workgroup's column tile is tn_begin + (tile % tiles_n): the P.V product (N = 1040 =
//      8 x 128 + 16) runs its 8 full column tiles on the straight-line kernel and the ninth on the ragged one; the score
//      product runs every column tile straight-line (its last one wastes <= 1 sub-tile of 80).
//  (2) ONE accumulator set per tile (the un-rescaled split needs no second chain; the three products of a tile are issued
//      in the order lo*hi, hi*lo, hi*hi): 166 VGPRs instead of 230.
// Measured (profiles/r05c_gemm_variants.txt, attention block of the B = 64 forward, same box): 26.5 ms (rounds 1-4 kernel) ->
// 26.2 (1) -> 25.85 (1 + 2): the exec-masked branches were NOT what holds the GEMMs at 0.34 of the matrix core's rate — LDS
// traffic is (both operands' hi and lo images: per CU and k-step 128 KB of ds_read_b128 + 64 KB of staging writes against
// 1536 cycles of matrix-pipe time).  The obvious alternative — B fragments straight from global memory (K-contiguous rows:
// a lane's fragment is 16 contiguous bytes), three-slot register ring, only A through LDS — was built and measured SLOWER:
// 31.7 ms (profiles/r05b_gemm_direct_b_negative.txt; 8 KB of 64-byte-segment loads per wave and k-step through the
// texture path cost more than the LDS round trip they replace); removed.
// MODE (lh_set_tuning key 17, A/B): 1 = staging registers ONE k-step ahead, double-buffered LDS, two workgroups per CU (the
// rounds 1-4 pipeline); 2 = staging registers TWO k-steps ahead (the loop unrolled by two: the loads of k-step ks + 2 are
// issued while ks is computed and land in a second register set), 3 = ONE LDS stage, two barriers per k-step, THREE
// workgroups per CU (166 VGPRs fit three waves per SIMD; 33 KB of LDS each).
template <int EPI, bool RAGGED, int MODE, bool EXTRA = false>
__device__ __forceinline__ void gemm_nt3_body(const _Float16* __restrict__ A, const _Float16* __restrict__ Bm,
                                              float* __restrict__ Cm, int M, int N, int K, int lda, int ldb, long strideA,
                                              long strideB, long imgA, long imgB, int ldc, long strideC, float scale, int nbatch,
                                              int tiles_m, int tiles_n, int tn_begin, int Bn, int T) {
    constexpr int NSTAGE = MODE == 3 ? 1 : 2;
    // EXTRA: the column tile is 128 + 16 wide (the P.V product's N = 1040 = 7 x 128 + 144: its last 16 columns ride on column
    // tile 7 instead of a ninth, ragged tile that re-read all of P for 1.5 % of the columns — 0.39 ms per call).  The 17th
    // sub-tile is split by ROWS so that every wave gets the same extra work: wave (wm, wn) computes rows wm*64 + (2 wn + ii)*16,
    // ii = 0, 1, of it (6 MFMAs per k-step on A fragments it has loaded anyway).
    // LDS image of an operand stage: four 16-byte k-block planes of RP rows, row r of plane kb in slot kb * RP + (r ^ 2 kb).
    // RP is a multiple of 16 slots (no padding) and the XOR does the de-conflicting for BOTH access shapes: ds_read_b128 is
    // served in the lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md, LDS) — a fragment read's group is rows
    // {0-3, 12-15} of plane g4 and rows {4-11} of plane g4 + 1: sixteen distinct slots mod 16 exactly when RP = 0 mod 16
    // (until r05i RP was padded to 130 for the staging stores: every fragment read 2-way conflicted, SQ_LDS_BANK_CONFLICT =
    // 0.333 of the LDS cycles, to the digit); ds_write_b128 is served in 8-lane groups = rows {s, s + 1} x planes 0..3
    // -> slots (s ^ {0, 2, 4, 6}) and ((s + 1) ^ {0, 2, 4, 6}) mod 8: all different.
    constexpr int RPA = 128, RPB = EXTRA ? 144 : 128, IMGA = 4 * RPA * 8, IMGB = 4 * RPB * 8;
    constexpr int STAGE_HALVES = 2 * IMGA + 2 * IMGB;
    __shared__ __attribute__((aligned(16))) _Float16 sm[NSTAGE][STAGE_HALVES];   // [stage][A hi | A lo | B hi | B lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware decode: consecutive workgroup ids go round-robin over the 8 XCDs; keep a batch on one XCD
    const int ntile = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int batch = (j / ntile) * 8 + xcd, tile = j % ntile;
    if (batch >= nbatch) return;
    const int m0 = (tile / tiles_n) * 128, n0 = (tn_begin + tile % tiles_n) * 128;
    const _Float16* Ab = A + (long)batch * strideA;
    const _Float16* Bb = Bm + (long)batch * strideB;

    // staging: 128 rows x 4 k-blocks of 16 bytes per image = 512 items, 2 per thread
    const int sr = tid >> 2, sb = tid & 3;
    const long a_off0 = (long)min(m0 + sr, M - 1) * lda + sb * 8, a_off1 = (long)min(m0 + sr + 64, M - 1) * lda + sb * 8;
    const long b_off0 = (long)min(n0 + sr, N - 1) * ldb + sb * 8, b_off1 = (long)min(n0 + sr + 64, N - 1) * ldb + sb * 8;
    const int srx = sr ^ (2 * sb), lx = l15 ^ (2 * g4);          // swizzled row within its 16-row group (staging / fragment side)
    const int s_idx0 = (sb * RPA + srx) * 8, s_idx1 = (sb * RPA + srx + 64) * 8;
    const int sb_idx0 = (sb * RPB + srx) * 8, sb_idx1 = (sb * RPB + srx + 64) * 8;
    // EXTRA: B rows 128 .. 143 of the tile = 64 more 16-byte items per image: threads 0 .. 63 (wave 0), one hi + one lo each
    const long bx_off = (long)min(n0 + 128 + sr, N - 1) * ldb + sb * 8;
    const int sbx_idx = (sb * RPB + 128 + srx) * 8;
    constexpr int NSET = MODE == 2 ? 2 : 1;
    f16x8 st[NSET][8], stx[NSET][2];
    auto fetch = [&](auto set_, int k0) {
        constexpr int q = decltype(set_)::value;
        st[q][0] = *reinterpret_cast<const f16x8*>(Ab + a_off0 + k0);        st[q][1] = *reinterpret_cast<const f16x8*>(Ab + a_off1 + k0);
        st[q][2] = *reinterpret_cast<const f16x8*>(Ab + imgA + a_off0 + k0); st[q][3] = *reinterpret_cast<const f16x8*>(Ab + imgA + a_off1 + k0);
        st[q][4] = *reinterpret_cast<const f16x8*>(Bb + b_off0 + k0);        st[q][5] = *reinterpret_cast<const f16x8*>(Bb + b_off1 + k0);
        st[q][6] = *reinterpret_cast<const f16x8*>(Bb + imgB + b_off0 + k0); st[q][7] = *reinterpret_cast<const f16x8*>(Bb + imgB + b_off1 + k0);
        if (EXTRA && wave == 0) {
            stx[q][0] = *reinterpret_cast<const f16x8*>(Bb + bx_off + k0);
            stx[q][1] = *reinterpret_cast<const f16x8*>(Bb + imgB + bx_off + k0);
        }
    };
    f32x4 amx[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 am[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) am[i][jn] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / 32;
    const int jn_live = __builtin_amdgcn_readfirstlane(min(4, max(0, (N - (n0 + wn * 64) + 15) / 16)));   // scalar, see above
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NSET - 1>;

    // one k-step: staged registers of set q -> LDS, barrier, refill the set with k-step ks + NSET, fragments + 48 MFMAs
    auto kstep = [&](int ks, auto set_) __attribute__((always_inline)) {
        constexpr int q = decltype(set_)::value;
        _Float16* buf = &sm[NSTAGE == 2 ? (ks & 1) : 0][0];
        _Float16* bufb = buf + 2 * IMGA;                          // B hi image, B lo image IMGB halves behind it
        if (NSTAGE == 1) __syncthreads();                         // every wave has read the previous k-step's fragments
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<f16x8*>(&buf[i * IMGA + s_idx0]) = st[q][2 * i];
            *reinterpret_cast<f16x8*>(&buf[i * IMGA + s_idx1]) = st[q][2 * i + 1];
            *reinterpret_cast<f16x8*>(&bufb[i * IMGB + sb_idx0]) = st[q][4 + 2 * i];
            *reinterpret_cast<f16x8*>(&bufb[i * IMGB + sb_idx1]) = st[q][4 + 2 * i + 1];
        }
        if (EXTRA && wave == 0) {
            *reinterpret_cast<f16x8*>(&bufb[sbx_idx]) = stx[q][0];
            *reinterpret_cast<f16x8*>(&bufb[IMGB + sbx_idx]) = stx[q][1];
        }
        __syncthreads();
        fetch(set_, min(ks + NSET, nk - 1) * 32);       // unconditional (the tail re-fetches the last k-step): a straight-line body keeps
                                                          // the compiler's vmcnt bookkeeping exact — behind a branch it drained every load
        __builtin_amdgcn_sched_barrier(0);                // the loads go out HERE (the scheduler otherwise sinks them below the MFMAs
                                                          // to recycle the staging registers as fragment registers: no prefetch left)
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int idx = (g4 * RPB + wn * 64 + jn * 16 + lx) * 8;
            bh[jn] = *reinterpret_cast<const f16x8*>(&bufb[idx]);
            bl[jn] = *reinterpret_cast<const f16x8*>(&bufb[IMGB + idx]);
        }
        f16x8 bxh, bxl;
        if (EXTRA) {
            const int idx = (g4 * RPB + 128 + lx) * 8;
            bxh = *reinterpret_cast<const f16x8*>(&bufb[idx]);
            bxl = *reinterpret_cast<const f16x8*>(&bufb[IMGB + idx]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = (g4 * RPA + wm * 64 + i * 16 + lx) * 8;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&buf[idx]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&buf[IMGA + idx]);
            // small terms first; the three products of one tile are four MFMAs apart (no back-to-back dependent MFMAs)
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn)
                    if (!RAGGED || jn < jn_live)
                        am[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? al : ah, p == 1 ? bl[jn] : bh[jn], am[i][jn], 0, 0, 0);
            if (EXTRA && (i >> 1) == wn) {                        // (scalar branch: wn comes from the scalar wave index)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    amx[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 0 ? al : ah, p == 1 ? bxl : bxh, amx[i & 1], 0, 0, 0);
            }
        }
    };
    fetch(S0{}, 0);
    if (NSET == 2) fetch(S1{}, min(1, nk - 1) * 32);          // (unconditional, like the refills)
    if (NSET == 2) {
        int ks = 0;
#pragma unroll 1
        for (; ks + 2 <= nk; ks += 2) {
            kstep(ks, S0{});
            kstep(ks + 1, S1{});
        }
        if (ks < nk) kstep(ks, S0{});
    } else {
#pragma unroll 1
        for (int ks = 0; ks < nk; ++ks) kstep(ks, S0{});
    }
    // epilogue: lane holds rows g4*4 + r, column l15 of each 16 x 16 tile -> 64-byte row segments
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int col = n0 + wn * 64 + jn * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + g4 * 4 + r;
                if (row < M && col < N) {
                    const float val = am[i][jn][r] * scale;
                    if (EPI == 0) {
                        Cm[(long)batch * strideC + (long)row * ldc + col] = val;
                    } else {                     // batch = h*Bn + b, row = frame, col = f*16 + v
                        const int hd = batch / Bn, b = batch % Bn;
                        Cm[(((long)b * T + row) * EF + (col >> 4)) * C + hd * VD + (col & 15)] = val;
                    }
                }
            }
        }
    if (EXTRA) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int col = n0 + 128 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + (2 * wn + ii) * 16 + g4 * 4 + r;
                if (row < M && col < N) {
                    const float val = amx[ii][r] * scale;
                    if (EPI == 0) {
                        Cm[(long)batch * strideC + (long)row * ldc + col] = val;
                    } else {
                        const int hd = batch / Bn, b = batch % Bn;
                        Cm[(((long)b * T + row) * EF + (col >> 4)) * C + hd * VD + (col & 15)] = val;
                    }
                }
            }
        }
    }
}
#define LH_GEMM_ARGS const _Float16* __restrict__ A, const _Float16* __restrict__ Bm, float* __restrict__ Cm, int M, int N, int K, \
    int lda, int ldb, long strideA, long strideB, long imgA, long imgB, int ldc, long strideC, float scale, int nbatch,   \
    int tiles_m, int tiles_n, int tn_begin, int Bn, int T
#define LH_GEMM_PASS A, Bm, Cm, M, N, K, lda, ldb, strideA, strideB, imgA, imgB, ldc, strideC, scale, nbatch, tiles_m, tiles_n, tn_begin, Bn, T
#if defined(LH_LEGACY)
template <int EPI, bool RAGGED, int MODE>
__global__ void __launch_bounds__(256, 2) k_gemm_nt3(LH_GEMM_ARGS) { gemm_nt3_body<EPI, RAGGED, MODE>(LH_GEMM_PASS); }
#endif
template <int EPI, bool RAGGED>
__global__ void __launch_bounds__(256, 3) k_gemm_nt3_occ3(LH_GEMM_ARGS) { gemm_nt3_body<EPI, RAGGED, 3>(LH_GEMM_PASS); }
// the 128 + 16 column tile: ~190 VGPRs (two more accumulator tiles, the 17th sub-tile's fragments and staging) = two
// workgroups per CU; one column tile in eight runs on it
template <int EPI>
__global__ void __launch_bounds__(256, 2) k_gemm_nt3_wide(LH_GEMM_ARGS) { gemm_nt3_body<EPI, false, 3, true>(LH_GEMM_PASS); }

// row softmax of the (already scaled) scores: one wave per row; P as fp16 hi/lo, keys zero-padded to Tp
__global__ void __launch_bounds__(256) k_emb_softmax(const float* __restrict__ sc, _Float16* __restrict__ p, long rows, int T,
                                                     int Tp, long img) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = sc + row * Tp;
    float mx = -3.0e38f;
    for (int i = lane; i < T; i += 64) mx = fmaxf(mx, sr[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int i = lane; i < T; i += 64) sum += __expf(sr[i] - mx);
    const float inv = 1.0f / wave_sum(sum);
    _Float16* ph = p + row * Tp;
    _Float16* pl = ph + img;
    for (int i = lane; i < Tp; i += 64) {
        const float x = i < T ? __expf(sr[i] - mx) * inv : 0.f;
        _Float16 h, l;
        split_hl(x, h, l);
        ph[i] = h;
        pl[i] = l;
    }
}

// The same softmax with the row held in registers (round 4): one wave per row, NC float4 per lane (Tp <= 256 NC), ONE pass
// over the scores (16-byte loads) and 8-byte hi / lo stores — the three-pass form above read every score three times with
// 4-byte loads and wrote 2-byte stores (1.46 ms per call at B = 64, 2.2 TB/s for 3.3 GB).  Same arithmetic and operation
// order per element (max, __expf(x - max), sum, product with 1 / sum).
template <int NC>
__global__ void __launch_bounds__(256) k_emb_softmax_reg(const float* __restrict__ sc, _Float16* __restrict__ p, long rows, int T,
                                                         int Tp, long img) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* sr = sc + row * Tp;
    float v[NC][4];
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i0 = (c * 64 + lane) * 4;
        float4 u = make_float4(-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f);
        if (i0 < Tp) u = *reinterpret_cast<const float4*>(&sr[i0]);          // (columns T .. Tp-1 of the buffer are never written)
        v[c][0] = i0 + 0 < T ? u.x : -3.0e38f; v[c][1] = i0 + 1 < T ? u.y : -3.0e38f;
        v[c][2] = i0 + 2 < T ? u.z : -3.0e38f; v[c][3] = i0 + 3 < T ? u.w : -3.0e38f;
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = fmaxf(mx, v[c][j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    // per-lane partial sums in the three-pass kernel's order: element i belongs to lane i % 64 there, to lane (i / 4) % 64 here —
    // the sum is re-associated (fp32, 1251 terms in [0, 1]: 1e-7 relative), the products are not
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = (c * 64 + lane) * 4 + j;
            v[c][j] = i < T ? __expf(v[c][j] - mx) : 0.f;
            sum += v[c][j];
        }
    const float inv = 1.0f / wave_sum(sum);
    _Float16* ph = p + row * Tp;
    _Float16* pl = ph + img;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i0 = (c * 64 + lane) * 4;
        if (i0 < Tp) {
            f16x4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 h, l;
                split_hl(v[c][j] * inv, h, l);
                h4[j] = h;
                l4[j] = l;
            }
            *reinterpret_cast<f16x4*>(&ph[i0]) = h4;
            *reinterpret_cast<f16x4*>(&pl[i0]) = l4;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Embedding head (tfgridnet_orig/tfgridnet.py:120-127): Linear(65*64 -> 256) on every frame, LayerNorm(256), mean
// over frames.  One workgroup = 64 frames of one utterance x all 256 outputs (8 waves x 2 column tiles x 4 row
// tiles), K = 4160 streamed in 64-wide chunks through a split-precision LDS image; the per-tile frame sums go to
// `part` and k_emb_head_mean adds them in a fixed order (bit-reproducible, no atomics).
// ---------------------------------------------------------------------------------------------------------------
constexpr int EH_K = EF * C;          // 4160, flat (f*64 + c) — the weight image is packed in that order
constexpr int EH_KS = EH_K / 32;      // 130
constexpr int EH_N = 256;
constexpr int EH_ROWS = 64;

__global__ void __launch_bounds__(512, 1) k_emb_head(const float* __restrict__ z, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ bias, const float* __restrict__ lnw,
                                                     const float* __restrict__ lnb, float* __restrict__ part, int T, int ntile) {
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * 4 * EH_ROWS * 8];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * 4 * EH_ROWS * 8];
    __shared__ float rs[8][EH_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    const int b = blockIdx.y, t0 = blockIdx.x * EH_ROWS;
    const float* zb = z + (long)b * T * EH_K;

    f32x4 am[4][2], ac[4][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const float bz = bias[wave * 32 + nt * 16 + l15];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { am[mt][nt] = f32x4{bz, bz, bz, bz}; ac[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const int r0 = tid >> 4, k4 = (tid & 15) * 4;                 // staging: rows r0 and r0 + 32, 4 features at k4
    const float* src0 = zb + (long)min(t0 + r0, T - 1) * EH_K + k4;
    const float* src1 = zb + (long)min(t0 + r0 + 32, T - 1) * EH_K + k4;
    float4 s0 = *reinterpret_cast<const float4*>(src0), s1 = *reinterpret_cast<const float4*>(src1);
#pragma unroll 1
    for (int kc = 0; kc < EH_KS / 2; ++kc) {
        __syncthreads();
        e_store_split4<EH_ROWS>(ahi, alo, r0, k4, s0.x, s0.y, s0.z, s0.w);
        e_store_split4<EH_ROWS>(ahi, alo, r0 + 32, k4, s1.x, s1.y, s1.z, s1.w);
        __syncthreads();
        if (kc + 1 < EH_KS / 2) {
            s0 = *reinterpret_cast<const float4*>(src0 + (kc + 1) * 64);
            s1 = *reinterpret_cast<const float4*>(src1 + (kc + 1) * 64);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 wh[2], wl[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const _Float16* p = w_pk + (((long)(wave * 2 + nt) * EH_KS + kc * 2 + ks) * 64 + lane) * 16;
                wh[nt] = *reinterpret_cast<const f16x8*>(p);
                wl[nt] = *reinterpret_cast<const f16x8*>(p + 8);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int idx = e_slot<EH_ROWS>(ks * 4 + g4, mt * 16 + l15);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    am[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[nt], am[mt][nt], 0, 0, 0);
                    ac[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[nt], ac[mt][nt], 0, 0, 0);
                    ac[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[nt], ac[mt][nt], 0, 0, 0);
                }
            }
        }
    }
    // ---- LayerNorm(256) per frame (two-pass), then the sum over this tile's valid frames
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) am[mt][nt][r] = am[mt][nt][r] + ac[mt][nt][r] * (1.0f / ESPLIT);
    float mean[4][4], rstd[4][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v;
                if (pass == 0) v = am[mt][0][r] + am[mt][1][r];
                else { const float d0 = am[mt][0][r] - mean[mt][r], d1 = am[mt][1][r] - mean[mt][r]; v = d0 * d0 + d1 * d1; }
                v = group16_sum(v);
                if (l15 == 0) rs[wave][mt * 16 + g4 * 4 + r] = v;
            }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = mt * 16 + g4 * 4 + r;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) v += rs[w][row];
                if (pass == 0) mean[mt][r] = v * (1.0f / EH_N);
                else rstd[mt][r] = rsqrtf(v * (1.0f / EH_N) + LN_EPS);
            }
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int col = wave * 32 + nt * 16 + l15;
        const float gw = lnw[col], gb = lnb[col];
        float sum = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (t0 + mt * 16 + g4 * 4 + r < T) sum += (am[mt][nt][r] - mean[mt][r]) * rstd[mt][r] * gw + gb;
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (g4 == 0) part[((long)b * ntile + blockIdx.x) * EH_N + col] = sum;
    }
}

__global__ void __launch_bounds__(256) k_emb_head_mean(const float* __restrict__ part, float* __restrict__ out, int T, int ntile) {
    const int b = blockIdx.x, col = threadIdx.x;
    float s = 0.f;
    for (int i = 0; i < ntile; ++i) s += part[((long)b * ntile + i) * EH_N + col];
    out[(long)b * EH_N + col] = s / (float)T;
}

}  // namespace lh

extern "C" int lh_emb_frontend(const float* x, float* inv_std, const float* wfb_pk, const float* wconv_pk,
                               const float* bconv, const float* gn_w, const float* gn_b, double* gn_part, float* z,
                               void* xsplit_next, int B, int T, int n_samples, lh_stream_t stream) {
    using namespace lh;
    if (!x || !inv_std || !wfb_pk || !wconv_pk || !bconv || !gn_w || !gn_b || !gn_part || !z || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    if (T != n_samples / EHOP + 1 || n_samples < ENFFT) return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_emb_std, dim3(B), dim3(ES_NTH), 0, st, x, inv_std, NMIC * n_samples);
    const int tiles = B * ((T + EM_TT - 1) / EM_TT);
    hipLaunchKernelGGL(k_emb_stft_conv, dim3(tiles < 256 ? tiles : 256), dim3(256), 0, st, x, inv_std, wfb_pk, wconv_pk,
                       bconv, z, gn_part, B, T, n_samples);
    hipLaunchKernelGGL(k_emb_gn_apply, dim3(64, B), dim3(256), 0, st, z, gn_part, gn_w, gn_b, (_Float16*)xsplit_next,
                       (long)B * T * EF, T);
    return check_launch();
}

#if defined(LH_LEGACY)
// One axis path of a GridNetBlock: inter = 0 along frequency (sequences = frames), 1 along time (sequences = bins).
//   x, out [B][T][65][64] (must not alias); wih_pk fp16 hi/lo image [32 ntiles][8 ksteps][64][16] of the folded,
//   column-permuted input weights of both directions; bih [512]; whh_pk [2][4][4][2][64][16]; wct_pk [4][16][64][16]
//   (ConvTranspose1d taps as [64 out] x [4*128]); bct [64]; xsplit scratch 2*B*T*65*64 fp16 (hi | lo images of the
//   channel-normalised input); gx scratch [nseq*P][512]; hbuf scratch [nseq*P][128]
extern "C" int lh_emb_axis(const float* x, const void* wih_pk, const float* bih, const void* whh_pk, const void* wct_pk,
                           const float* bct, void* xsplit, float* gx, float* hbuf, float* out, int B, int T, int inter,
                           lh_stream_t stream) {
    using namespace lh;
    if (!x || !wih_pk || !bih || !whh_pk || !wct_pk || !bct || !xsplit || !gx || !hbuf || !out || B <= 0 || T < EKS ||
        x == out)
        return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nseq = inter ? B * EF : B * T;
    const int P = (inter ? T : EF) - (EKS - 1);
    const int gtiles = nseq * ((P + 63) / 64);
    const long rows = (long)B * T * EF;
    const int ctiles = (int)((rows + 31) / 32);
    const int gxgrid = gtiles < 128 ? gtiles : 128;                 // x 4 column chunks = 2 workgroups per CU
    const long lnb = (rows + 15) / 16;
    hipLaunchKernelGGL(k_emb_lnsplit, dim3((unsigned)(lnb < 4096 ? lnb : 4096)), dim3(256), 0, st, x, (_Float16*)xsplit, rows);
    if (inter) {
        hipLaunchKernelGGL((k_emb_gx<true>), dim3(gxgrid, 4), dim3(256), 0, st, (const _Float16*)xsplit,
                           (const _Float16*)wih_pk, bih, gx, nseq, P, T, rows);
        hipLaunchKernelGGL(k_emb_lstm, dim3((nseq + 15) / 16, 2), dim3(256), 0, st, gx, (const _Float16*)whh_pk, hbuf, nseq, P);
        hipLaunchKernelGGL((k_emb_convt_res<true>), dim3(ctiles < 512 ? ctiles : 512), dim3(256), 0, st, hbuf,
                           (const _Float16*)wct_pk, bct, x, out, rows, P, T);
    } else {
        hipLaunchKernelGGL((k_emb_gx<false>), dim3(gxgrid, 4), dim3(256), 0, st, (const _Float16*)xsplit,
                           (const _Float16*)wih_pk, bih, gx, nseq, P, T, rows);
        hipLaunchKernelGGL(k_emb_lstm, dim3((nseq + 15) / 16, 2), dim3(256), 0, st, gx, (const _Float16*)whh_pk, hbuf, nseq, P);
        hipLaunchKernelGGL((k_emb_convt_res<false>), dim3(ctiles < 512 ? ctiles : 512), dim3(256), 0, st, hbuf,
                           (const _Float16*)wct_pk, bct, x, out, rows, P, T);
    }
    return check_launch();
}

#endif  // LH_LEGACY

#if defined(ER_TRACE)
extern "C" int lh_probe_er_trace_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::er_trace_buf), sizeof(lh::er_trace_buf)) == hipSuccess ? 0 : 1;
}
#endif
namespace lh {
static int g_rec_prio = 0;              // lh_set_tuning key 16: issue priority for k_emb_rec's on-chain MFMAs (0 = off)
static int g_gemm_v = 3;                // lh_set_tuning key 17: k_gemm_nt3 pipeline 1 / 2 / 3 (see gemm_nt3_body); 0 = the rounds 1-4 k_gemm_nt (-DLH_LEGACY builds only)
int emb_set(int key, int value) {
    if ((key != 16 && key != 17) || value < 0 || value > (key == 17 ? 3 : 1)) return LH_ERR_ARG;
#if !defined(LH_LEGACY)
    if (key == 17 && value != 3) return LH_ERR_UNSUPPORTED;     // the other GEMM pipelines exist in lab builds only
#endif
    if (key == 16) g_rec_prio = value;
    else g_gemm_v = value;
    return LH_OK;
}
}  // namespace lh

// One axis path, round-4 form (see k_emb_rec): LayerNorm + split (k_emb_lnsplit) -> k_emb_rec (input GEMM + recurrence,
// both directions in one launch) -> k_emb_convt2 (+ residual).  No gate pre-activation buffer.
//   wrec_pk fp16 [2 dirs][8 waves][40 fragments][64 lanes][8] (embed_net.py pack_rec); brec [2][256] in (unit, gate) order,
//   pre-scaled; wct_pk, bct as lh_emb_axis; xsplit scratch 2*B*T*65*64 fp16; hsplit scratch 2*nseq*P*128 fp16
//   have_xsplit: xsplit already holds LayerNorm(x) split (written by the previous call with emit_split, or by
//   lh_emb_attn_block's xsplit_next) — the normalisation launch is skipped;  emit_split: after the recurrence has consumed
//   xsplit, the transposed-conv kernel overwrites it with LayerNorm(out) split, for the next axis call
extern "C" int lh_emb_axis_fused(const float* x, const void* wrec_pk, const float* brec, const void* wct_pk, const float* bct,
                                 void* xsplit, void* hsplit, float* out, int B, int T, int inter, int have_xsplit,
                                 int emit_split, lh_stream_t stream) {
    using namespace lh;
    if (!x || !wrec_pk || !brec || !wct_pk || !bct || !xsplit || !hsplit || !out || B <= 0 || T < EKS || x == out)
        return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nseq = inter ? B * EF : B * T;
    const int Lp = inter ? T : EF;
    const int P = Lp - (EKS - 1);
    const long rows = (long)B * T * EF;
    const long lnb = (rows + 15) / 16;
    const int rt = inter ? CtShape<true>::RT : CtShape<false>::RT;
    const int ctiles = nseq * ((Lp + rt - 1) / rt);
    if (!have_xsplit)
        hipLaunchKernelGGL(k_emb_lnsplit, dim3((unsigned)(lnb < 4096 ? lnb : 4096)), dim3(256), 0, st, x, (_Float16*)xsplit, rows);
    _Float16* xs_next = emit_split ? (_Float16*)xsplit : nullptr;
    if (inter) {
        hipLaunchKernelGGL((k_emb_rec<true>), dim3((nseq + 15) / 16, 2), dim3(ER_NT), 0, st, (const _Float16*)xsplit,
                           (const _Float16*)wrec_pk, brec, (_Float16*)hsplit, nseq, P, T, rows, g_rec_prio);
        hipLaunchKernelGGL((k_emb_convt2<true>), dim3(ctiles < 512 ? ctiles : 512), dim3(256), 0, st, (const _Float16*)hsplit,
                           (const _Float16*)wct_pk, bct, x, out, xs_next, rows, nseq, P, T);
    } else {
        hipLaunchKernelGGL((k_emb_rec<false>), dim3((nseq + 15) / 16, 2), dim3(ER_NT), 0, st, (const _Float16*)xsplit,
                           (const _Float16*)wrec_pk, brec, (_Float16*)hsplit, nseq, P, T, rows, g_rec_prio);
        hipLaunchKernelGGL((k_emb_convt2<false>), dim3(ctiles < 512 ? ctiles : 512), dim3(256), 0, st, (const _Float16*)hsplit,
                           (const _Float16*)wct_pk, bct, x, out, xs_next, rows, nseq, P, T);
    }
    return check_launch();
}

// The inter-axis path for SMALL batches (round 6): as lh_emb_axis_fused(inter = 1), with the recurrence on one workgroup per
// (sequence, direction) — k_emb_inter_mv — instead of 16-sequence tiles.  Same results to fp32 rounding (the input half is the
// same split-precision product in another summation order, the recurrent half an fp32 mat-vec instead of three fp16 MFMAs).
//   wih_pk, bih   the `_pack_axis` images of embed_net.py (natural tap order, columns (direction, unit, gate), unscaled)
//   whh           fp32 [2][256][64], row 4 unit + gate
extern "C" int lh_emb_axis_mv(const float* x, const void* wih_pk, const float* bih, const float* whh, const void* wct_pk,
                              const float* bct, void* xsplit, void* hsplit, float* out, int B, int T, int have_xsplit,
                              int emit_split, lh_stream_t stream) {
    using namespace lh;
    if (!x || !wih_pk || !bih || !whh || !wct_pk || !bct || !xsplit || !hsplit || !out || B <= 0 || T < EKS || x == out)
        return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nseq = B * EF, P = T - (EKS - 1);
    const long rows = (long)B * T * EF;
    const long lnb = (rows + 15) / 16;
    const int ctiles = nseq * ((T + CtShape<true>::RT - 1) / CtShape<true>::RT);
    if (!have_xsplit)
        hipLaunchKernelGGL(k_emb_lnsplit, dim3((unsigned)(lnb < 4096 ? lnb : 4096)), dim3(256), 0, st, x, (_Float16*)xsplit, rows);
    if (2 * nseq <= 256)          // one workgroup per CU at most: the eight-wave form (its extra waves halve the input GEMM's time)
        hipLaunchKernelGGL(k_emb_inter_mv<IS_NT>, dim3(2 * nseq), dim3(IS_NT), 0, st, (const _Float16*)xsplit, (const _Float16*)wih_pk,
                           bih, whh, (_Float16*)hsplit, nseq, P, T, rows);
    else
        hipLaunchKernelGGL(k_emb_inter_mv<IS_NR>, dim3(2 * nseq), dim3(IS_NR), 0, st, (const _Float16*)xsplit, (const _Float16*)wih_pk,
                           bih, whh, (_Float16*)hsplit, nseq, P, T, rows);
    hipLaunchKernelGGL((k_emb_convt2<true>), dim3(ctiles < 512 ? ctiles : 512), dim3(256), 0, st, (const _Float16*)hsplit,
                       (const _Float16*)wct_pk, bct, x, out, emit_split ? (_Float16*)xsplit : nullptr, rows, nseq, P, T);
    return check_launch();
}

// Attention branch of one GridNetBlock: Q/K/V frame kernel, V transpose, score GEMM, softmax, P.V GEMM with fused
// head merge, projection + LayerNorm + residual.  Tp = T rounded up to 64.
//   y2, out [B][T][65][64]; merged scratch [B][T][65][64]
//   q, k  scratch fp16 [2][4B][T][544]   (hi | lo images);  v scratch fp32 [4B][T][1040]
//   vt    scratch fp16 [2][4B][1040][Tp]; sc scratch fp32 [4B][T][Tp]; p scratch fp16 [2][4B][T][Tp]
//   wqkv_pk fp16 hi/lo image [8][2][64][16] of the stacked 1x1-conv weights [128 x 64]; bqkv, slopes [128];
//   lnq/lnk [4][520], lnv [4][1040] affine in flat (f*d + c) order; wproj_pk [4][2][64][16]; bproj [64]; slope_p [1];
//   lnp_w/b [4160] in flat (f*64 + c) order
extern "C" int lh_emb_attn_block(const float* y2, const void* wqkv_pk, const float* bqkv, const float* slopes,
                                 const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                                 const float* lnv_w, const float* lnv_b, const void* wproj_pk, const float* bproj,
                                 const float* slope_p, const float* lnp_w, const float* lnp_b, void* q, void* k, float* v,
                                 void* vt, float* sc, void* p, float* merged, float* out, void* xsplit_next, int B, int T,
                                 lh_stream_t stream) {
    using namespace lh;
    if (!y2 || !wqkv_pk || !bqkv || !slopes || !lnq_w || !lnq_b || !lnk_w || !lnk_b || !lnv_w || !lnv_b || !wproj_pk ||
        !bproj || !slope_p || !lnp_w || !lnp_b || !q || !k || !v || !vt || !sc || !p || !merged || !out || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int nframes = B * T, nb = NH * B, Tp = (T + 63) / 64 * 64;
    hipLaunchKernelGGL(k_emb_qkv, dim3(nframes < 512 ? nframes : 512), dim3(256), 0, st, y2, (const _Float16*)wqkv_pk, bqkv,
                       slopes, lnq_w, lnq_b, lnk_w, lnk_b, lnv_w, lnv_b, (_Float16*)q, (_Float16*)k, v, B, T);
    const long img_qk = (long)nb * T * EQP, img_vt = (long)nb * EDV * Tp, img_p = (long)nb * T * Tp;
    hipLaunchKernelGGL(k_emb_vt, dim3(Tp / 64, (EDV + 63) / 64, nb), dim3(256), 0, st, v, (_Float16*)vt, T, Tp, img_vt);
    const int nb8 = (nb + 7) / 8 * 8;
    const int tm = (T + 127) / 128;
#if defined(LH_LEGACY)
    if (g_gemm_v == 0)        // lab builds: the rounds 1-4 kernel (lh_set_tuning(17, 0))
        hipLaunchKernelGGL((k_gemm_nt<0>), dim3(nb8 * tm * tm), dim3(256), 0, st, (const _Float16*)q, (const _Float16*)k, sc, T, T,
                           EQP, EQP, EQP, (long)T * EQP, (long)T * EQP, img_qk, img_qk, Tp, (long)T * Tp,
                           1.0f / sqrtf((float)EDQK), nb, tm, tm, B, T);
    else
#endif
    // scores: every column tile on the straight-line kernel (the last one wastes <= 1 sub-tile of 80)
#define LH_QK_LAUNCH(KERNEL) hipLaunchKernelGGL(KERNEL, dim3(nb8 * tm * tm), dim3(256), 0, st, (const _Float16*)q, (const _Float16*)k, sc, T, T, \
    EQP, EQP, EQP, (long)T * EQP, (long)T * EQP, img_qk, img_qk, Tp, (long)T * Tp, 1.0f / sqrtf((float)EDQK), nb, tm, tm, 0, B, T)
#if defined(LH_LEGACY)       // lab builds keep the measured-slower pipelines selectable (profiles/r05d_gemm_pipelines.txt)
    if (g_gemm_v == 2) LH_QK_LAUNCH((k_gemm_nt3<0, false, 2>));
    else if (g_gemm_v == 1) LH_QK_LAUNCH((k_gemm_nt3<0, false, 1>));
    else
#endif
    LH_QK_LAUNCH((k_gemm_nt3_occ3<0, false>));
#undef LH_QK_LAUNCH
    const long rows = (long)nb * T;
    {
        const dim3 sg((unsigned)((rows + 3) / 4)), sb(256);
        const int nc = (Tp + 255) / 256;             // float4 per lane with the row in registers (clips up to ~8 s: nc <= 8)
#define LH_SOFTMAX_REG(NC_) hipLaunchKernelGGL((k_emb_softmax_reg<NC_>), sg, sb, 0, st, sc, (_Float16*)p, rows, T, Tp, img_p)
        switch (nc) {
            case 1: LH_SOFTMAX_REG(1); break;
            case 2: LH_SOFTMAX_REG(2); break;
            case 3: LH_SOFTMAX_REG(3); break;
            case 4: LH_SOFTMAX_REG(4); break;
            case 5: LH_SOFTMAX_REG(5); break;
            case 6: LH_SOFTMAX_REG(6); break;
            case 7: LH_SOFTMAX_REG(7); break;
            case 8: LH_SOFTMAX_REG(8); break;
            default: hipLaunchKernelGGL(k_emb_softmax, sg, sb, 0, st, sc, (_Float16*)p, rows, T, Tp, img_p);   // longer clips: three passes
        }
#undef LH_SOFTMAX_REG
    }
    const int tn = (EDV + 127) / 128;
#if defined(LH_LEGACY)
    if (g_gemm_v == 0)
        hipLaunchKernelGGL((k_gemm_nt<1>), dim3(nb8 * tm * tn), dim3(256), 0, st, (const _Float16*)p, (const _Float16*)vt, merged, T,
                           EDV, Tp, Tp, Tp, (long)T * Tp, (long)EDV * Tp, img_p, img_vt, 0, 0L, 1.0f, nb, tm, tn, B, T);
    else
#endif
    {
        const int tn_full = EDV / 128;            // 8 full column tiles on the straight-line kernel, the ragged ninth on its own
#define LH_PV_LAUNCH(KERNEL, NT, TB) hipLaunchKernelGGL(KERNEL, dim3(nb8 * tm * (NT)), dim3(256), 0, st, (const _Float16*)p, (const _Float16*)vt, \
    merged, T, EDV, Tp, Tp, Tp, (long)T * Tp, (long)EDV * Tp, img_p, img_vt, 0, 0L, 1.0f, nb, tm, (NT), (TB), B, T)
        static_assert(EDV % 128 == 16 && EDV / 128 >= 1, "the wide last column tile takes exactly 16 columns behind the full tiles");
#if defined(LH_LEGACY)
        if (g_gemm_v == 1 || g_gemm_v == 2) {
            if (g_gemm_v == 2) LH_PV_LAUNCH((k_gemm_nt3<1, false, 2>), tn_full, 0);
            else LH_PV_LAUNCH((k_gemm_nt3<1, false, 1>), tn_full, 0);
            if (tn > tn_full) LH_PV_LAUNCH((k_gemm_nt3<1, true, 1>), tn - tn_full, tn_full);
        } else
#endif
        {
            // three workgroups per CU; the 16 columns behind the last full tile ride on it (k_gemm_nt3_wide) — no ragged launch
            if (tn_full > 1) LH_PV_LAUNCH((k_gemm_nt3_occ3<1, false>), tn_full - 1, 0);
            LH_PV_LAUNCH((k_gemm_nt3_wide<1>), 1, tn_full - 1);
        }
#undef LH_PV_LAUNCH
    }
    hipLaunchKernelGGL(k_emb_proj, dim3(nframes < 512 ? nframes : 512), dim3(256), 0, st, merged, (const _Float16*)wproj_pk,
                       bproj, slope_p, lnp_w, lnp_b, y2, out, (_Float16*)xsplit_next, nframes);
    return check_launch();
}

// Embedding head: z [B][T][65][64] -> emb [B][256].  w_pk: fp16 hi/lo image [16][130][64][16] of the Linear weight
// with its input features reordered to (f*64 + c); part: scratch [B][ceil(T/64)][256].
extern "C" int lh_emb_head(const float* z, const void* w_pk, const float* bias, const float* ln_w, const float* ln_b,
                           float* part, float* emb, int B, int T, lh_stream_t stream) {
    using namespace lh;
    if (!z || !w_pk || !bias || !ln_w || !ln_b || !part || !emb || B <= 0 || T <= 0) return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int ntile = (T + EH_ROWS - 1) / EH_ROWS;
    hipLaunchKernelGGL(k_emb_head, dim3(ntile, B), dim3(512), 0, st, z, (const _Float16*)w_pk, bias, ln_w, ln_b, part, T, ntile);
    hipLaunchKernelGGL(k_emb_head_mean, dim3(B), dim3(256), 0, st, part, emb, T, ntile);
    return check_launch();
}
