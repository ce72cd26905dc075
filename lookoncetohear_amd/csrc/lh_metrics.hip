// On-device evaluation metrics of the reference test loop (reference src/ts_hear_test.py:139-146): per utterance
// SI-SNR(output, target) and SI-SNRi = SI-SNR(output, target) - SI-SNR(mixture, target), averaged over the two
// channels, plus the cosine similarity of the enrollment embedding — so a sharded eval moves 32 bytes per rank
// instead of copying [B, 2, 80000] waveforms to the host.
//
// torchmetrics' scale_invariant_signal_noise_ratio = zero-mean SI-SDR:
//     alpha = (<p,t> + eps) / (<t,t> + eps);  10 log10((|alpha t|^2 + eps) / (|alpha t - p|^2 + eps)),  eps = fp32 eps
// on mean-removed p, t.  The moments are accumulated in fp64 (the noise energy is a difference of nearly equal
// sums), streamed once: 3 waveforms in, 8 doubles per (utterance, channel, chunk) out.
#include "lh_common.h"

namespace lh {

constexpr int MT_CHUNKS = 16;     // workgroups per (utterance, channel)
constexpr int MT_NM = 8;          // sum p, t, m, pp, tt, mm, pt, mt

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// grid (MT_CHUNKS, B*2), block 256
__global__ void __launch_bounds__(256) k_metric_moments(const float* __restrict__ pred, const float* __restrict__ tgt,
                                                        const float* __restrict__ mix, double* __restrict__ part, int n) {
    __shared__ double red[4][MT_NM];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    const long base = (long)blockIdx.y * n;
    const int per = ((n + MT_CHUNKS - 1) / MT_CHUNKS + 3) & ~3;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    double s[MT_NM];
#pragma unroll
    for (int k = 0; k < MT_NM; ++k) s[k] = 0.0;
    const bool vec = ((base | lo) & 3) == 0;
    for (int i = lo + tid * 4; i < hi; i += 256 * 4) {
        float p[4], t[4], m[4];
        if (vec && i + 3 < hi) {
            const float4 p4 = *reinterpret_cast<const float4*>(&pred[base + i]);
            const float4 t4 = *reinterpret_cast<const float4*>(&tgt[base + i]);
            const float4 m4 = *reinterpret_cast<const float4*>(&mix[base + i]);
            p[0] = p4.x; p[1] = p4.y; p[2] = p4.z; p[3] = p4.w;
            t[0] = t4.x; t[1] = t4.y; t[2] = t4.z; t[3] = t4.w;
            m[0] = m4.x; m[1] = m4.y; m[2] = m4.z; m[3] = m4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = i + j < hi;
                p[j] = ok ? pred[base + i + j] : 0.f;
                t[j] = ok ? tgt[base + i + j] : 0.f;
                m[j] = ok ? mix[base + i + j] : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double pd = p[j], td = t[j], md = m[j];
            s[0] += pd; s[1] += td; s[2] += md;
            s[3] += pd * pd; s[4] += td * td; s[5] += md * md;
            s[6] += pd * td; s[7] += md * td;
        }
    }
#pragma unroll
    for (int k = 0; k < MT_NM; ++k) {
        const double v = wave_sum_f64(s[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < MT_NM)
        part[((long)blockIdx.y * MT_CHUNKS + blockIdx.x) * MT_NM + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

__device__ __forceinline__ double si_snr_from_moments(double sp, double st, double spp, double stt, double spt, double n) {
    const double eps = 1.1920928955078125e-07;                // torch.finfo(float32).eps
    const double mp = sp / n, mt = st / n;
    const double pt = spt - n * mp * mt, tt = stt - n * mt * mt, pp = spp - n * mp * mp;
    const double alpha = (pt + eps) / (tt + eps);
    const double sig = alpha * alpha * tt;
    const double noise = sig - 2.0 * alpha * pt + pp;
    return 10.0 * log10((sig + eps) / (fmax(noise, 0.0) + eps));
}

// per-utterance rows [B][3] = (output_sisnr, si_snr_i, embedding_sim).  One wave per utterance, four utterances per
// workgroup, grid ceil(B / 4): lanes 0..15 add up the 16 chunk partials of their (channel, moment), all 64 lanes share the
// cosine similarity.  (Rounds 1-4 ran ONE workgroup whose four waves walked B / 4 utterances each — four software fp64
// log10 chains per utterance back to back: 57 us per call at B = 32, 0.8 % of the batch-32 step for 96 numbers.)
__global__ void __launch_bounds__(256) k_metric_finish(const double* __restrict__ part, const float* __restrict__ emb,
                                                       const float* __restrict__ emb_gt, float* __restrict__ rows,
                                                       int B, int n, int edim) {
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
    double* rows64 = const_cast<double*>(part) + (long)B * 2 * MT_CHUNKS * MT_NM;     // [B][3] tail of the scratch
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    double m = 0.0;                                           // lane = ch * 8 + k (< 16): moment k of channel ch
    if (lane < 2 * MT_NM) {
        double pc[MT_CHUNKS];
#pragma unroll
        for (int c = 0; c < MT_CHUNKS; ++c) pc[c] = part[(((long)b * 2 + (lane >> 3)) * MT_CHUNKS + c) * MT_NM + (lane & 7)];
#pragma unroll
        for (int c = 0; c < MT_CHUNKS; ++c) m += pc[c];       // chunk order: the sum does not depend on the launch shape
    }
    double out_sisnr = 0.0, snr_i = 0.0;
    for (int ch = 0; ch < 2; ++ch) {
        double mm[MT_NM];
#pragma unroll
        for (int k = 0; k < MT_NM; ++k) mm[k] = __shfl(m, ch * MT_NM + k);
        const double so = si_snr_from_moments(mm[0], mm[1], mm[3], mm[4], mm[6], (double)n);
        const double sm = si_snr_from_moments(mm[2], mm[1], mm[5], mm[4], mm[7], (double)n);
        out_sisnr += 0.5 * so;
        snr_i += 0.5 * (so - sm);
    }
    double ab = 0.0, aa = 0.0, bb = 0.0;
    for (int i = lane; i < edim; i += 64) {
        const double x = emb[(long)b * edim + i], yv = emb_gt[(long)b * edim + i];
        ab += x * yv; aa += x * x; bb += yv * yv;
    }
    ab = wave_sum_f64(ab); aa = wave_sum_f64(aa); bb = wave_sum_f64(bb);
    const double cosv = ab / (fmax(sqrt(aa), 1e-8) * fmax(sqrt(bb), 1e-8));         // F.cosine_similarity, eps 1e-8
    if (lane == 0) {
        rows[b * 3 + 0] = (float)out_sisnr;
        rows[b * 3 + 1] = (float)snr_i;
        rows[b * 3 + 2] = (float)cosv;
        rows64[b * 3 + 0] = snr_i; rows64[b * 3 + 1] = out_sisnr; rows64[b * 3 + 2] = cosv;
    }
}

// sums[4] (fp64) = [sum si_snr_i, sum output_sisnr, sum embedding_sim, B]: one wave, lane k adds column k of the fp64 rows in
// utterance order (sequential: bit-reproducible, and the same order as the single-thread loop of rounds 1-4)
__global__ void __launch_bounds__(64) k_metric_total(const double* __restrict__ part, double* __restrict__ sums, int B) {
    const double* rows64 = part + (long)B * 2 * MT_CHUNKS * MT_NM;
    const int k = threadIdx.x;
    if (k < 3) {
        double a = 0.0;
        for (int b = 0; b < B; ++b) a += rows64[b * 3 + k];
        sums[k] = a;
    } else if (k == 3) {
        sums[3] = (double)B;
    }
}

}  // namespace lh

extern "C" int lh_metric_sums(const float* outputs, const float* target, const float* mixture, const float* emb,
                              const float* emb_gt, double* scratch, float* rows, double* sums, int B, int n_samples,
                              int emb_dim, lh_stream_t stream) {
    using namespace lh;
    if (!outputs || !target || !mixture || !emb || !emb_gt || !scratch || !rows || !sums || B <= 0 || n_samples <= 0 ||
        emb_dim <= 0)
        return LH_ERR_ARG;
    hipLaunchKernelGGL(k_metric_moments, dim3(MT_CHUNKS, B * 2), dim3(256), 0, (hipStream_t)stream, outputs, target,
                       mixture, scratch, n_samples);
    hipLaunchKernelGGL(k_metric_finish, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, scratch, emb, emb_gt, rows, B,
                       n_samples, emb_dim);
    hipLaunchKernelGGL(k_metric_total, dim3(1), dim3(64), 0, (hipStream_t)stream, scratch, sums, B);
    return check_launch();
}
