// Binaural rendering — the step before the separator: every mono source of an utterance is
// convolved with a 2-ear impulse response and the results are mixed, as the reference's data pipeline does on the
// CPU with scipy per source (reference src/datasets/multi_ch_simulator.py:40-61 `_convolve`:
// `convolve(src, rir[ear])[:len(src)]`; src/datasets/MixLibriSpeechNoisyEnrollNorm.py:176-202: noise scale, peak
// normalisation, mixture sum, target selection).
//
//   k_fir_causal   y[n] = sum_{k < Lh} h[k] x[n - k], n < N (x[<0] = 0), one (utterance, source, ear) channel per
//                  grid.y.  Plain fp32 VALU FMAs on purpose: the output has only 2 columns per source, so a
//                  Toeplitz-GEMM on the fp32 MFMA wastes half of its lanes on structural zeros and lands at the same
//                  78 TFLOP/s as the vector unit.  Each thread owns 8 consecutive outputs and keeps a sliding
//                  15-sample window of x in registers: 8 LDS reads and 8 scalar filter taps feed 64 FMAs.
//   k_fft_conv     the same convolution by overlap-save FFT for 1024 <= Lh <= 4097 taps (BRIR-length responses)
//   k_mix_peak     peak of |sum of sources| per utterance (order-independent: atomicMax on the float bit pattern)
//   k_mix_apply    events / norm (when norm > 1), mixture = ((e0 + e1) + e2) + noise in the reference's order, target
#include "lh_common.h"

namespace lh {

constexpr int FIR_R = 8;                       // outputs per thread
constexpr int FIR_TILE = 256 * FIR_R;          // 2048 outputs per workgroup
constexpr int FIR_KT = 2048;                   // taps per LDS stage (multiple of FIR_R)
constexpr int FIR_SPAN = FIR_TILE + FIR_KT + FIR_R;      // samples staged per stage
__device__ __forceinline__ int fir_slot(int p) { return p + (p >> 3); }   // 1 pad word per 8: lane stride 8 -> 9 banks

// grid (ceil(N / FIR_TILE), channels); x [nch/2][N] (both ears of a source share it), h [nch][Lh], y [nch][N]
__global__ void __launch_bounds__(256) k_fir_causal(const float* __restrict__ x, const float* __restrict__ h,
                                                    const float* __restrict__ gain, float* __restrict__ y, int N, int Lh) {
    __shared__ float xs[FIR_SPAN + FIR_SPAN / 8 + 8];
    const int tid = threadIdx.x;
    const int ch = blockIdx.y, n0 = blockIdx.x * FIR_TILE;
    const float* xc = x + (long)(ch >> 1) * N;
    const float* hc = h + (long)ch * Lh;
    float acc[FIR_R];
#pragma unroll
    for (int r = 0; r < FIR_R; ++r) acc[r] = 0.f;
    // taps beyond n0 + FIR_TILE - 1 only ever meet x[<0] = 0: skip them
    const int lh_eff = min(Lh, n0 + FIR_TILE);
    for (int k0 = 0; k0 < lh_eff; k0 += FIR_KT) {
        // stage x[lo .. lo + FIR_SPAN - 1], lo = n0 - k0 - FIR_KT - FIR_R + 1 (zeros outside [0, N))
        const int lo = n0 - k0 - FIR_KT - FIR_R + 1;
        __syncthreads();
        for (int i = tid; i < FIR_SPAN; i += 256) {
            const int p = lo + i;
            xs[fir_slot(i)] = (p >= 0 && p < N) ? xc[p] : 0.f;
        }
        __syncthreads();
        // thread's outputs n = n0 + tid*R + r; tap kt = k0 + k + kk needs x[n - kt] = staged sample
        // (tid*R + FIR_KT - k) + (R-1) + r - kk.  tid*R + FIR_KT - k is a multiple of 8, so the 8 samples a chunk of 8
        // taps adds to the window are 8 consecutive LDS words at 9*(tid + (FIR_KT - k)/8); the other 7 window samples
        // are the previous chunk's (two register sets swap roles: no moves)
        const int kend = min(FIR_KT, lh_eff - k0);
        float wa[FIR_R], wb[FIR_R];
        {
            const float* wp = xs + 9 * (tid + FIR_KT / 8 + 1);
#pragma unroll
            for (int j = 0; j < FIR_R - 1; ++j) wb[j] = wp[j];
            wb[FIR_R - 1] = 0.f;
        }
        auto chunk = [&](float (&lo)[FIR_R], const float (&hi)[FIR_R], int k) {
            const float* wp = xs + 9 * (tid + (FIR_KT - k) / 8);
#pragma unroll
            for (int j = 0; j < FIR_R; ++j) lo[j] = wp[j];
            const int kt0 = k0 + k;
            float hk[FIR_R];
            if (kt0 + FIR_R <= Lh) {                               // wave-uniform: one 32-byte scalar load
#pragma unroll
                for (int kk = 0; kk < FIR_R; ++kk) hk[kk] = hc[kt0 + kk];
            } else {
#pragma unroll
                for (int kk = 0; kk < FIR_R; ++kk) hk[kk] = kt0 + kk < Lh ? hc[kt0 + kk] : 0.f;
            }
#pragma unroll
            for (int kk = 0; kk < FIR_R; ++kk)
#pragma unroll
                for (int r = 0; r < FIR_R; ++r) {
                    const int idx = FIR_R - 1 + r - kk;            // compile-time after unrolling
                    acc[r] = fmaf(hk[kk], idx >= FIR_R ? hi[idx - FIR_R] : lo[idx], acc[r]);
                }
        };
        for (int k = 0; k < kend; k += 2 * FIR_R) {
            chunk(wa, wb, k);
            if (k + FIR_R < kend) chunk(wb, wa, k + FIR_R);
        }
    }
    const float g = gain[ch >> 1];
    float* yc = y + (long)ch * N;
    const int n = n0 + tid * FIR_R;
    if (n + FIR_R <= N && ((N & 3) == 0)) {
        *reinterpret_cast<float4*>(&yc[n]) = make_float4(acc[0] * g, acc[1] * g, acc[2] * g, acc[3] * g);
        *reinterpret_cast<float4*>(&yc[n + 4]) = make_float4(acc[4] * g, acc[5] * g, acc[6] * g, acc[7] * g);
    } else {
#pragma unroll
        for (int r = 0; r < FIR_R; ++r)
            if (n + r < N) yc[n + r] = acc[r] * g;
    }
}

// The overlap-save FFT path (responses of 1024..4097 taps) lives in lh_render_fft.hip: its complex arithmetic turns into
// packed fp32 with crossed operand selects under the vectorisers, the instruction form that is unsafe on this chip
// (build.py), so that file is compiled without them while this one keeps v_pk_fma_f32 for the direct-form FIR.
constexpr int FC_LOG = 13, FC_F = 1 << FC_LOG;    // transform size
constexpr int FC_L = FC_F / 2;                    // outputs per block (Lh - 1 <= FC_F - FC_L)
constexpr int FC_LH_MIN = 1024, FC_LH_MAX = FC_F - FC_L + 1;
int launch_fft_conv(const float* x, const float* h, const float* gain, float* y, int N, int Lh, int rows, hipStream_t st);

// sum of the S1 rendered sources in the reference's order: ((e0 + e1) + ...) + noise (the last row)
__device__ __forceinline__ float mix_sum(const float* __restrict__ ev, long stride, int S1, long i, float inv, bool scale) {
    float s = 0.f;
    for (int k = 0; k < S1; ++k) {
        float e = ev[k * stride + i];
        if (scale) e = e / inv;                     // `multi_ch_events[i] /= norm_factor`: IEEE division, like torch
        s = k == 0 ? e : s + e;
    }
    return s;
}

// grid (chunks, B): peak[b] = max |mixture| over both ears (bit pattern of a non-negative float orders like uint)
__global__ void __launch_bounds__(256) k_mix_peak(const float* __restrict__ ev, unsigned* __restrict__ peak_bits, int S1, int N) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long stride = 2L * N;
    const float* evb = ev + (long)b * S1 * stride;
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < stride; i += (long)gridDim.x * 256)
        m = fmaxf(m, fabsf(mix_sum(evb, stride, S1, i, 1.f, false)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&peak_bits[b], __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

__global__ void __launch_bounds__(256) k_mix_apply(float* __restrict__ ev, const unsigned* __restrict__ peak_bits,
                                                   const int* __restrict__ tgt_idx, float* __restrict__ mixture,
                                                   float* __restrict__ target, int S1, int N) {
    const int b = blockIdx.y;
    const long stride = 2L * N;
    float* evb = ev + (long)b * S1 * stride;
    const float nf = __uint_as_float(peak_bits[b]);
    const bool scale = nf > 1.0f;
    const int tg = tgt_idx[b];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < stride; i += (long)gridDim.x * 256) {
        mixture[(long)b * stride + i] = mix_sum(evb, stride, S1, i, nf, scale);
        const float t = evb[tg * stride + i];
        target[(long)b * stride + i] = scale ? t / nf : t;
    }
}

}  // namespace lh

// Renders B utterances of S1 mono rows each (sources first, the noise bed LAST) to two ears and mixes them.
//   src      [B][S1][N]      mono signals
//   rir      [B][S1][2][Lh]  impulse response per row and ear (zero-pad shorter ones to Lh)
//   gain     [B][S1]         applied after the convolution: 1 for sources, `noise_scale` for the noise row
//   tgt_idx  [B] int32       row returned as `target`
//   events   [B][S1][2][N]   out: rendered rows BEFORE peak normalisation (scratch for the mix)
//   peak     [B] uint32      out: bit pattern of the fp32 peak of |mixture| (the reference's `norm_factor`)
//   mixture, target [B][2][N] out: divided by the peak when it exceeds 1
extern "C" int lh_render_binaural(const float* src, const float* rir, const float* gain, const int* tgt_idx, float* events,
                                  unsigned* peak, float* mixture, float* target, int B, int S1, int N, int Lh,
                                  lh_stream_t stream) {
    using namespace lh;
    if (!src || !rir || !gain || !tgt_idx || !events || !peak || !mixture || !target || B <= 0 || S1 <= 0 || N <= 0 || Lh <= 0)
        return LH_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(peak, 0, sizeof(unsigned) * B, st) != hipSuccess) return LH_ERR_LAUNCH;
    if (Lh >= FC_LH_MIN && Lh <= FC_LH_MAX) {
        // room-length responses: overlap-save FFT convolution, 5 output blocks of 4096 samples per workgroup
        if (launch_fft_conv(src, rir, gain, events, N, Lh, B * S1, st) != LH_OK) return LH_ERR_LAUNCH;
    } else {
        hipLaunchKernelGGL(k_fir_causal, dim3((N + FIR_TILE - 1) / FIR_TILE, B * S1 * 2), dim3(256), 0, st, src, rir, gain,
                           events, N, Lh);
    }
    const int chunks = (int)((2L * N + 256 * 8 - 1) / (256 * 8));
    hipLaunchKernelGGL(k_mix_peak, dim3(chunks, B), dim3(256), 0, st, events, peak, S1, N);
    hipLaunchKernelGGL(k_mix_apply, dim3(chunks, B), dim3(256), 0, st, events, peak, tgt_idx, mixture, target, S1, N);
    return check_launch();
}
