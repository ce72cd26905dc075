// Pointwise (per time-frequency unit) channel contractions on fp32 MFMA with fused epilogues:
//   k_linear_res   out = res + W h + b                       (intra_linear / inter_linear + residual)
//   k_qkv_proj_ln  Q/K/V = LN_(f,e)(PReLU(W y + b)) per head  (attn_conv_Q/K/V)
//   k_proj_ln_res  out = (y2 + LN_(f,c)(PReLU(W m + b))) * gain   (attn_concat_proj + residual + speaker gain)
// All three stage a [rows x K] activation tile in LDS (coalesced 256-byte rows), keep the weight matrix in
// VGPRs as v_mfma_f32_16x16x4_f32 B fragments, and run their LayerNorm/activation epilogue out of LDS so
// every activation byte crosses HBM once per stage (SURVEY.md §8d "algorithmic bytes").
#include "lh_common.h"

namespace lh {

// ------------------------------------------------------------------------------------------------------
// out[r][0:64] = res[r][0:64] + bias + sum_k h[r][k] * W[o][k]       K in {64, 128}
// persistent grid-stride over 64-row tiles; weights loaded once per workgroup.
// ------------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) k_linear_res(const float* __restrict__ h, const float* __restrict__ w_pk,
                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                    float* __restrict__ out, int rows) {
    constexpr int KC = K / 4;            // floats per k-chunk (one chunk per 16-lane group)
    constexpr int KP = KC + 4;           // padded LDS row
    constexpr int CP = C + 4;
    __shared__ __attribute__((aligned(16))) float as[4 * 64 * KP];
    __shared__ __attribute__((aligned(16))) float cs[4 * 16 * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    float wreg[4][KC];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < KC; ++ks) wreg[nt][ks] = w_pk[(nt * KC + ks) * 64 + lane];
    float bz[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bz[nt] = bias[nt * 16 + l15];

    const int ntiles = (rows + 63) / 64;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = (long)tile * 64;
        // stage 64 rows x K floats (float4 granules, a row is contiguous)
        for (int e = tid; e < 64 * (K / 4); e += 256) {
            const int rl = e / (K / 4), qq = e % (K / 4);
            const long r = min(r0 + rl, (long)rows - 1);
            const float4 v = *reinterpret_cast<const float4*>(&h[r * K + qq * 4]);
            const int chunk = qq / (KC / 4), j4 = qq % (KC / 4);
            *reinterpret_cast<float4*>(&as[(chunk * 64 + rl) * KP + j4 * 4]) = v;
        }
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{bz[nt], bz[nt], bz[nt], bz[nt]};
        const float* arow = &as[(g4 * 64 + wave * 16 + l15) * KP];
#pragma unroll
        for (int qq = 0; qq < KC / 4; ++qq) {
            const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wreg[nt][qq * 4 + j], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(wave * 16 + g4 * 4 + r) * CP + nt * 16 + l15] = acc[nt][r];
        __syncthreads();
        // residual add + coalesced store: each wave writes its own 16 rows
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i, rr = e >> 4, qq = e & 15;
            const long r = r0 + wave * 16 + rr;
            if (r < rows) {
                const float4 cv = *reinterpret_cast<const float4*>(&cs[(wave * 16 + rr) * CP + qq * 4]);
                const float4 rv = *reinterpret_cast<const float4*>(&res[r * C + qq * 4]);
                *reinterpret_cast<float4*>(&out[r * C + qq * 4]) =
                    make_float4(cv.x + rv.x, cv.y + rv.y, cv.z + rv.z, cv.w + rv.w);
            }
        }
        // the next iteration's staging barrier orders these LDS reads before cs is rewritten
    }
}

// ------------------------------------------------------------------------------------------------------
// shared piece: stage one frame [97 x 64] into the 4-chunk LDS image used as MFMA A operand (K = 64)
// ------------------------------------------------------------------------------------------------------
constexpr int FR_MT = 7;                  // 7 M tiles cover 97 rows (112)
constexpr int FR_KP = 20;                 // 16-float k-chunk + 4 pad

__device__ __forceinline__ void stage_frame(const float* __restrict__ src, float* as, int tid) {
    for (int e = tid; e < NF * 16; e += 256) {
        const int rl = e >> 4, qq = e & 15;
        const float4 v = *reinterpret_cast<const float4*>(&src[rl * C + qq * 4]);
        *reinterpret_cast<float4*>(&as[((qq >> 2) * (FR_MT * 16) + rl) * FR_KP + (qq & 3) * 4]) = v;
    }
    for (int e = tid; e < (FR_MT * 16 - NF) * 16; e += 256) {      // zero the 15 padding rows
        const int rl = NF + (e >> 4), qq = e & 15;
        *reinterpret_cast<float4*>(&as[((qq >> 2) * (FR_MT * 16) + rl) * FR_KP + (qq & 3) * 4]) =
            make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// One wave: LayerNorm over the NF*D values ys[f][col0 + e] (flat index f*D + e), affine per flat index, written
// to dst[0..ld) (columns >= NF*D are zero padding).  D is compile-time so the index split is shifts / mul-shift.
template <int D>
__device__ __forceinline__ void ln_head(const float* ys, int yp, int col0, const float* __restrict__ gw,
                                        const float* __restrict__ gb, float* __restrict__ dst, int ld, int lane) {
    constexpr int N = NF * D;
    float s = 0.f;
    for (int i = lane; i < N; i += 64) s += ys[(i / D) * yp + col0 + (i % D)];
    const float mean = wave_sum(s) * (1.0f / N);
    float v = 0.f;
    for (int i = lane; i < N; i += 64) { const float dv = ys[(i / D) * yp + col0 + (i % D)] - mean; v += dv * dv; }
    const float rstd = rsqrtf(wave_sum(v) * (1.0f / N) + LN_EPS);
    for (int i = lane; i < ld; i += 64) {
        float o = 0.f;
        if (i < N) o = (ys[(i / D) * yp + col0 + (i % D)] - mean) * rstd * gw[i] + gb[i];
        dst[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------------
// Q/K/V projection + PReLU + per-head LayerNorm over (f,e); persistent workgroups, grid-stride over frames
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_qkv_proj_ln(const float* __restrict__ y, const float* __restrict__ w_pk,
                                                     const float* __restrict__ bias, const float* __restrict__ slopes,
                                                     const float* __restrict__ lnq_w, const float* __restrict__ lnq_b,
                                                     const float* __restrict__ lnk_w, const float* __restrict__ lnk_b,
                                                     const float* __restrict__ lnv_w, const float* __restrict__ lnv_b,
                                                     float* __restrict__ q, float* __restrict__ kx,
                                                     float* __restrict__ vx, int T, int nframes) {
    constexpr int NT = NQKV / 16;         // 7 N tiles
    constexpr int YP = NQKV + 1;          // 113: odd stride -> conflict-free column walks in the LN phase
    __shared__ __attribute__((aligned(16))) float as[4 * FR_MT * 16 * FR_KP];
    __shared__ float ys[NF * YP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    float wreg[NT][16];                   // weights loaded once per persistent workgroup
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wreg[nt][ks] = w_pk[(nt * 16 + ks) * 64 + lane];
    const float sq = slopes[0], sk = slopes[1], sv = slopes[2];

  for (int fr = blockIdx.x; fr < nframes; fr += gridDim.x) {      // grid-stride over frames (b*T + t)
    const int b = fr / T, t = fr % T;
    stage_frame(y + (long)fr * NF * C, as, tid);
    __syncthreads();                      // also orders the previous frame's LDS reads of `ys`

    for (int m = wave; m < FR_MT; m += 4) {
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float bz = bias[nt * 16 + l15];
            acc[nt] = f32x4{bz, bz, bz, bz};
        }
        const float* arow = &as[(g4 * (FR_MT * 16) + m * 16 + l15) * FR_KP];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wreg[nt][qq * 4 + j], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + l15;
            const float a = col < NH * E ? sq : (col < 2 * NH * E ? sk : sv);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + g4 * 4 + r;
                if (row < NF) ys[row * YP + col] = prelu_f(acc[nt][r], a);
            }
        }
    }
    __syncthreads();

    // per-head LayerNorm: wave w normalises head w of Q, K and V; flat index = f*D + e (F-major, e-minor)
    const int hd = wave;
    const long bh = (long)b * NH + hd;
    ln_head<E>(ys, YP, hd * E, lnq_w, lnq_b, q + (bh * T + t) * LDQK, LDQK, lane);
    ln_head<E>(ys, YP, NH * E + hd * E, lnk_w, lnk_b, kx + (bh * (T + HIST) + HIST + t) * LDQK, LDQK, lane);
    ln_head<VD>(ys, YP, 2 * NH * E + hd * VD, lnv_w, lnv_b, vx + (bh * (T + HIST) + HIST + t) * DV, DV, lane);
  }
}

// ------------------------------------------------------------------------------------------------------
// attn_concat_proj + LN over (f,c) + residual (+ speaker gain);  grid (T, B), one frame per workgroup
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_proj_ln_res(const float* __restrict__ merged, const float* __restrict__ w_pk,
                                                     const float* __restrict__ bias, const float* __restrict__ slope,
                                                     const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                     const float* __restrict__ y2, const float* __restrict__ gain,
                                                     float* __restrict__ out, int T, int nframes) {
    constexpr int YP = C + 4;
    __shared__ __attribute__((aligned(16))) float as[4 * FR_MT * 16 * FR_KP];
    __shared__ __attribute__((aligned(16))) float ys[NF * YP];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    float wreg[4][16];                    // weights loaded once per persistent workgroup
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wreg[nt][ks] = w_pk[(nt * 16 + ks) * 64 + lane];
    const float a = slope[0];
    // LayerNorm affine of this thread's fixed float4 slots, resident for all frames of the persistent loop
    constexpr int NSLOT = (NF * C / 4 + 255) / 256;      // 7
    float4 pw[NSLOT], pb[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int i = min(tid + 256 * k, NF * C / 4 - 1);
        pw[k] = *reinterpret_cast<const float4*>(&lnw[i * 4]);
        pb[k] = *reinterpret_cast<const float4*>(&lnb[i * 4]);
    }

  for (int fidx = blockIdx.x; fidx < nframes; fidx += gridDim.x) {     // grid-stride over frames (b*T + t)
    const int b = fidx / T;
    const long fr = (long)fidx * NF * C;
    stage_frame(merged + fr, as, tid);
    __syncthreads();                      // also orders the previous frame's LDS reads of `ys`

    for (int m = wave; m < FR_MT; m += 4) {
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const float bz = bias[nt * 16 + l15];
            acc[nt] = f32x4{bz, bz, bz, bz};
        }
        const float* arow = &as[(g4 * (FR_MT * 16) + m * 16 + l15) * FR_KP];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const float4 a4 = *reinterpret_cast<const float4*>(arow + qq * 4);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wreg[nt][qq * 4 + j], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + g4 * 4 + r;
                if (row < NF) ys[row * YP + nt * 16 + l15] = prelu_f(acc[nt][r], a);
            }
    }
    __syncthreads();

    // joint LayerNorm over all 97*64 values of the frame (flat index f*64 + c), float4 granules
    constexpr int N = NF * C, N4 = N / 4;
    float s = 0.f;
    for (int i = tid; i < N4; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(&ys[(i >> 4) * YP + (i & 15) * 4]);
        s += v.x + v.y + v.z + v.w;
    }
    const float mean = block_sum_256(s, red) * (1.0f / N);
    float vs = 0.f;
    for (int i = tid; i < N4; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(&ys[(i >> 4) * YP + (i & 15) * 4]);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        vs += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    const float rstd = rsqrtf(block_sum_256(vs, red) * (1.0f / N) + LN_EPS);
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int i = tid + 256 * k;
        if (i >= N4) break;
        const float4 v = *reinterpret_cast<const float4*>(&ys[(i >> 4) * YP + (i & 15) * 4]);
        const float4 gw = pw[k], gb = pb[k];
        const float4 rv = *reinterpret_cast<const float4*>(&y2[fr + i * 4]);
        float4 o;
        o.x = rv.x + (v.x - mean) * rstd * gw.x + gb.x;
        o.y = rv.y + (v.y - mean) * rstd * gw.y + gb.y;
        o.z = rv.z + (v.z - mean) * rstd * gw.z + gb.z;
        o.w = rv.w + (v.w - mean) * rstd * gw.w + gb.w;
        if (gain) {
            const float4 gv = *reinterpret_cast<const float4*>(&gain[(long)b * N + i * 4]);
            o.x *= gv.x; o.y *= gv.y; o.z *= gv.z; o.w *= gv.w;
        }
        *reinterpret_cast<float4*>(&out[fr + i * 4]) = o;
    }
  }
}

}  // namespace lh

extern "C" int lh_linear_res(const float* h, const float* w_pk, const float* bias, const float* res, float* out,
                             int rows, int K, lh_stream_t stream) {
    using namespace lh;
    if (!h || !w_pk || !bias || !res || !out || rows <= 0) return LH_ERR_ARG;
    const int ntiles = (rows + 63) / 64;
    const int grid = ntiles < 1024 ? ntiles : 1024;
    if (K == 128)
        hipLaunchKernelGGL((k_linear_res<128>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w_pk, bias, res, out, rows);
    else if (K == 64)
        hipLaunchKernelGGL((k_linear_res<64>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, w_pk, bias, res, out, rows);
    else
        return LH_ERR_UNSUPPORTED;
    return check_launch();
}

extern "C" int lh_qkv_proj_ln(const float* y, const float* w_pk, const float* bias, const float* slopes,
                              const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                              const float* lnv_w, const float* lnv_b, float* q, float* kx, float* vx, int B, int T,
                              lh_stream_t stream) {
    using namespace lh;
    if (!y || !w_pk || !bias || !slopes || !lnq_w || !lnq_b || !lnk_w || !lnk_b || !lnv_w || !lnv_b || !q || !kx ||
        !vx || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    const int nframes = B * T;
    hipLaunchKernelGGL(k_qkv_proj_ln, dim3(nframes < 512 ? nframes : 512), dim3(256), 0, (hipStream_t)stream, y, w_pk,
                       bias, slopes, lnq_w, lnq_b, lnk_w, lnk_b, lnv_w, lnv_b, q, kx, vx, T, nframes);
    return check_launch();
}

extern "C" int lh_proj_ln_res(const float* merged, const float* w_pk, const float* bias, const float* slope,
                              const float* ln_w, const float* ln_b, const float* y2, const float* gain, float* out,
                              int B, int T, lh_stream_t stream) {
    using namespace lh;
    if (!merged || !w_pk || !bias || !slope || !ln_w || !ln_b || !y2 || !out || B <= 0 || T <= 0) return LH_ERR_ARG;
    const int nframes = B * T;
    hipLaunchKernelGGL(k_proj_ln_res, dim3(nframes < 512 ? nframes : 512), dim3(256), 0, (hipStream_t)stream, merged,
                       w_pk, bias, slope, ln_w, ln_b, y2, gain, out, T, nframes);
    return check_launch();
}
