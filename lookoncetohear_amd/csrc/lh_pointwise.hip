// Pointwise (per time-frequency unit) channel contractions with fused epilogues:
//   k_linear_res   out = res + W h + b                            (intra_linear / inter_linear + residual)
//   k_qkv_proj_ln  Q/K/V = LN_(f,e)(PReLU(W y + b)) per head       (attn_conv_Q/K/V)
//   k_proj_ln_res  out = (y2 + LN_(f,c)(PReLU(W m + b))) * gain    (attn_concat_proj + residual + speaker gain)
//
// These are HBM-streaming stages: every activation byte should cross HBM once per stage, so
//   * workgroups are persistent (grid-stride over row tiles / frames) and keep their weights in VGPRs;
//   * global loads of a tile are all issued before the first dependent LDS write (register staging; hipcc does
//     not unroll a load->ds_write loop by itself and otherwise serialises one HBM round trip per float4), and the
//     next frame is prefetched into registers while the current one is processed;
//   * the contraction runs on split-precision fp16 MFMA ("f16x3": v = hi + 2^-11 lo, products hi*hi + 2^-11
//     (hi*lo + lo*hi) on v_mfma_f32_16x16x32_f16, ~22 mantissa bits, see lh_lstm.hip) so the matrix pipe costs
//     ~1/5 of fp32 MFMA and stays hidden behind the memory stream;
//   * each WAVE owns output-column tiles (not row tiles): its weight fragments are 16-32 registers instead of
//     112-128, which keeps 2-3 workgroups resident per CU for latency hiding;
//   * LayerNorm / PReLU / residual epilogues run out of LDS with compile-time index algebra.
#include "lh_common.h"

namespace lh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr float PW_SPLIT = 2048.0f;

// A-operand LDS image for v_mfma_f32_16x16x32_f16: [block = kstep*4 + 16-lane group][row slot][8 halves].
// Reads: a lane's 16 bytes of consecutive rows are consecutive 16-byte slots.  Writes come row-major from the
// coalesced global loads (16 lanes = one row = 8 blocks x 2 halves), and the block stride is a multiple of
// 128 bytes, so the row slot is XOR-swizzled with the block index: the 8 blocks of one row land in 8 different
// slots (conflict-free ds_write_b64), while within any ds_read_b128 lane group the XOR only permutes rows
// inside aligned groups of 4 (or swaps the two halves of the group), which keeps the reads conflict-free.
template <int RP>
__device__ __forceinline__ int a_slot(int blk, int row) { return (blk * RP + (row ^ (blk & 7))) * 8; }
template <int RP>
__device__ __forceinline__ int a_index(int row, int k) {
    return a_slot<RP>((k >> 5) * 4 + ((k >> 3) & 3), row) + (k & 7);
}

template <int RP>
__device__ __forceinline__ void store_split4(_Float16* ahi, _Float16* alo, int row, int k0, float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    f16x4 h4, l4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 h = (_Float16)x[i];
        h4[i] = h;
        l4[i] = (_Float16)((x[i] - (float)h) * PW_SPLIT);
    }
    const int idx = a_index<RP>(row, k0);
    *reinterpret_cast<f16x4*>(&ahi[idx]) = h4;
    *reinterpret_cast<f16x4*>(&alo[idx]) = l4;
}

// bias + A[m-tile rows] * W[n-tile]  for K = 32*KS; wh/wl = hi/lo B fragments of this wave's n-tile
template <int RP, int KS>
__device__ __forceinline__ f32x4 mma_tile(const _Float16* ahi, const _Float16* alo, int m, int g4, int l15,
                                          const f16x8 (&wh)[KS], const f16x8 (&wl)[KS], float bias) {
    f32x4 am = f32x4{bias, bias, bias, bias}, ac = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int idx = a_slot<RP>(ks * 4 + g4, m * 16 + l15);
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
        const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[ks], am, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[ks], ac, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[ks], ac, 0, 0, 0);
    }
    return am + ac * (1.0f / PW_SPLIT);
}

// weight image: [n-tile][kstep][lane][hi 8 | lo 8] fp16 (weights.py: pack_linear_f16x3)
template <int KS>
__device__ __forceinline__ void load_w(const _Float16* __restrict__ w_pk, int nt, int lane, f16x8 (&wh)[KS],
                                       f16x8 (&wl)[KS]) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const _Float16* p = w_pk + ((long)(nt * KS + ks) * 64 + lane) * 16;
        wh[ks] = *reinterpret_cast<const f16x8*>(p);
        wl[ks] = *reinterpret_cast<const f16x8*>(p + 8);
    }
}

// ------------------------------------------------------------------------------------------------------
// out[r][0:64] = res[r][0:64] + bias + sum_k h[r][k] * W[o][k]       K in {64, 128}; 64-row tiles
// ------------------------------------------------------------------------------------------------------
template <int K>
__global__ void __launch_bounds__(256) k_linear_res(const float* __restrict__ h, const _Float16* __restrict__ w_pk,
                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                    float* __restrict__ out, int rows) {
    constexpr int KS = K / 32, RP = 64, CP = C + 4;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[KS * 4 * RP * 8];
    __shared__ __attribute__((aligned(16))) _Float16 alo[KS * 4 * RP * 8];
    __shared__ __attribute__((aligned(16))) float cs[RP * CP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    f16x8 wh[KS], wl[KS];
    load_w<KS>(w_pk, wave, lane, wh, wl);              // wave w owns output columns 16w .. 16w+15
    const float bz = bias[wave * 16 + l15];

    const int ntiles = (rows + RP - 1) / RP;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long r0 = (long)tile * RP;
        constexpr int NLD = RP * (K / 4) / 256;
        float4 stg[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {                // all global loads of the tile in flight
            const int e = tid + 256 * i;
            const long r = min(r0 + e / (K / 4), (long)rows - 1);
            stg[i] = *reinterpret_cast<const float4*>(&h[r * K + (e % (K / 4)) * 4]);
        }
        // residual rows: issue the loads now, they land while the MFMAs run
        float4 rv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            const long r = min(r0 + (e >> 4), (long)rows - 1);
            rv[i] = *reinterpret_cast<const float4*>(&res[r * C + (e & 15) * 4]);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = tid + 256 * i;
            store_split4<RP>(ahi, alo, e / (K / 4), (e % (K / 4)) * 4, stg[i]);
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < RP / 16; ++m) {
            const f32x4 acc = mma_tile<RP, KS>(ahi, alo, m, g4, l15, wh, wl, bz);
#pragma unroll
            for (int r = 0; r < 4; ++r) cs[(m * 16 + g4 * 4 + r) * CP + wave * 16 + l15] = acc[r];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, rr = e >> 4, qq = e & 15;
            const long r = r0 + rr;
            if (r < rows) {
                const float4 cv = *reinterpret_cast<const float4*>(&cs[rr * CP + qq * 4]);
                *reinterpret_cast<float4*>(&out[r * C + qq * 4]) =
                    make_float4(cv.x + rv[i].x, cv.y + rv[i].y, cv.z + rv[i].z, cv.w + rv[i].w);
            }
        }
        // the next tile's staging barrier orders these cs reads before cs is rewritten
    }
}

// ------------------------------------------------------------------------------------------------------
// frame staging shared by the two frame kernels: [97 x 64] fp32 -> hi/lo fp16 A image (rows padded to 112)
// ------------------------------------------------------------------------------------------------------
constexpr int FR_RP = 112;
constexpr int FR_A = 2 * 4 * FR_RP * 8;    // halves per image (K = 64 -> 2 k-steps)
constexpr int FR_NLD = (NF * 16 + 255) / 256;

__device__ __forceinline__ void frame_load(const float* __restrict__ src, int tid, float4 (&stg)[FR_NLD]) {
#pragma unroll
    for (int i = 0; i < FR_NLD; ++i) {
        const int e = min(tid + 256 * i, NF * 16 - 1);
        stg[i] = *reinterpret_cast<const float4*>(&src[(e >> 4) * C + (e & 15) * 4]);
    }
}
// same, from the attention kernel's head-major frame [4 heads][97][16]: channel c = head*16 + v
__device__ __forceinline__ void frame_load_heads(const float* __restrict__ src, int tid, float4 (&stg)[FR_NLD]) {
#pragma unroll
    for (int i = 0; i < FR_NLD; ++i) {
        const int e = min(tid + 256 * i, NF * 16 - 1);
        stg[i] = *reinterpret_cast<const float4*>(&src[((e & 15) >> 2) * DV + (e >> 4) * VD + (e & 3) * 4]);
    }
}
__device__ __forceinline__ void frame_store(_Float16* ahi, _Float16* alo, int tid, const float4 (&stg)[FR_NLD]) {
#pragma unroll
    for (int i = 0; i < FR_NLD; ++i) {
        const int e = tid + 256 * i;
        if (e < NF * 16) store_split4<FR_RP>(ahi, alo, e >> 4, (e & 15) * 4, stg[i]);
    }
}
// rows 97..111 of the image feed only accumulator rows that are dropped, but must hold finite numbers
__device__ __forceinline__ void frame_zero_pad(_Float16* ahi, _Float16* alo, int tid) {
    for (int e = tid; e < (FR_RP - NF) * 16; e += 256)
        store_split4<FR_RP>(ahi, alo, NF + (e >> 4), (e & 15) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
}

// One wave: LayerNorm over the NF*D values ys[f][col0 + e] (flat index i = f*D + e), affine per flat index, written
// to dst[0..ld) (columns >= NF*D are zero padding).  Slot k of a lane is i = lane + 64k; out-of-range slots are
// predicated (not clamped) so that for D = 16 every address is `base + k * constant` (immediate offsets, no
// per-slot address registers kept live across the persistent frame loop).
template <int D>
__device__ __forceinline__ void ln_head(const float* ys, int yp, int col0, const float* __restrict__ gw,
                                        const float* __restrict__ gb, float* __restrict__ dst, int ld, int lane) {
    constexpr int N = NF * D;
    constexpr int IT = (N + 63) / 64;              // 10 (Q/K) or 25 (V) slots per lane
    auto at = [&](int k) -> float {
        const int i = lane + 64 * k;
        return i < N ? ys[(i / D) * yp + col0 + (i % D)] : 0.f;
    };
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) s += at(k);
    const float mean = wave_sum(s) * (1.0f / N);
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const float dv = at(k) - mean;
        if (lane + 64 * k < N) v += dv * dv;
    }
    const float rstd = rsqrtf(wave_sum(v) * (1.0f / N) + LN_EPS);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int i = lane + 64 * k;
        if (i < N) dst[i] = (at(k) - mean) * rstd * gw[i] + gb[i];
        else if (i < ld) dst[i] = 0.f;             // pad columns (q/kx) stay 0
    }
}

// ------------------------------------------------------------------------------------------------------
// Q/K/V projection + PReLU + per-head LayerNorm over (f,e); persistent workgroups, grid-stride over frames
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_qkv_proj_ln(const float* __restrict__ y, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ bias, const float* __restrict__ slopes,
                                                     const float* __restrict__ lnq_w, const float* __restrict__ lnq_b,
                                                     const float* __restrict__ lnk_w, const float* __restrict__ lnk_b,
                                                     const float* __restrict__ lnv_w, const float* __restrict__ lnv_b,
                                                     float* __restrict__ q, float* __restrict__ kx,
                                                     float* __restrict__ vx, int T, int nframes) {
    constexpr int YP = NQKV + 1;          // 113: odd stride -> conflict-free column walks in the LN phase
    __shared__ __attribute__((aligned(16))) _Float16 ahi[FR_A];
    __shared__ __attribute__((aligned(16))) _Float16 alo[FR_A];
    __shared__ float ys[NF * YP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    // wave w owns output-column tiles w and w+4 (of 7): Q|K|V columns 16w.. and 64+16w..
    f16x8 wh0[2], wl0[2], wh1[2], wl1[2];
    load_w<2>(w_pk, wave, lane, wh0, wl0);
    const bool two = wave + 4 < NQKV / 16;
    load_w<2>(w_pk, two ? wave + 4 : wave, lane, wh1, wl1);
    const int c0 = wave * 16 + l15, c1 = c0 + 64;
    const float bz0 = bias[c0], bz1 = bias[two ? c1 : c0];
    const float sq = slopes[0], sk = slopes[1], sv = slopes[2];
    const float a0 = c0 < NH * E ? sq : (c0 < 2 * NH * E ? sk : sv);      // PReLU slope of column c0; c1 is always V

    frame_zero_pad(ahi, alo, tid);
    float4 stg[FR_NLD];
    if ((int)blockIdx.x < nframes) frame_load(y + (long)blockIdx.x * NF * C, tid, stg);
    for (int fr = blockIdx.x; fr < nframes; fr += gridDim.x) {      // grid-stride over frames (b*T + t)
        const int b = fr / T, t = fr % T;
        frame_store(ahi, alo, tid, stg);
        __syncthreads();                      // image complete; also orders the previous frame's reads of `ys`
        if (fr + (int)gridDim.x < nframes) frame_load(y + (long)(fr + gridDim.x) * NF * C, tid, stg);   // prefetch

#pragma unroll 1
        for (int m = 0; m < FR_RP / 16; ++m) {
            const f32x4 r0 = mma_tile<FR_RP, 2>(ahi, alo, m, g4, l15, wh0, wl0, bz0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + g4 * 4 + r;
                if (row < NF) ys[row * YP + c0] = prelu_f(r0[r], a0);
            }
            if (two) {
                const f32x4 r1 = mma_tile<FR_RP, 2>(ahi, alo, m, g4, l15, wh1, wl1, bz1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m * 16 + g4 * 4 + r;
                    if (row < NF) ys[row * YP + c1] = prelu_f(r1[r], sv);
                }
            }
        }
        __syncthreads();

        // per-head LayerNorm: wave w normalises head w of Q, K and V; flat index = f*D + e (F-major, e-minor)
        const int hd = wave;
        const long bh = (long)b * NH + hd;
        // `fr >> 30` is always 0 but ties the lane index to the loop variable: without it LICM hoists ~45 slot
        // addresses per head out of the persistent frame loop and the kernel spills
        const int ln = lane + (fr >> 30);
        ln_head<E>(ys, YP, hd * E, lnq_w, lnq_b, q + (bh * T + t) * LDQK, LDQK, ln);
        ln_head<E>(ys, YP, NH * E + hd * E, lnk_w, lnk_b, kx + (bh * (T + HIST) + HIST + t) * LDQK, LDQK, ln);
        ln_head<VD>(ys, YP, 2 * NH * E + hd * VD, lnv_w, lnv_b, vx + (bh * (T + HIST) + HIST + t) * DV, DV, ln);
    }
}

// ------------------------------------------------------------------------------------------------------
// attn_concat_proj + LN over (f,c) + residual (+ speaker gain); persistent, grid-stride over frames
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) k_proj_ln_res(const float* __restrict__ merged, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ bias, const float* __restrict__ slope,
                                                     const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                     const float* __restrict__ y2, const float* __restrict__ gain,
                                                     float* __restrict__ out, int T, int nframes) {
    constexpr int YP = C + 4;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[FR_A];
    __shared__ __attribute__((aligned(16))) _Float16 alo[FR_A];
    __shared__ __attribute__((aligned(16))) float ys[NF * YP];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;

    f16x8 wh[2], wl[2];
    load_w<2>(w_pk, wave, lane, wh, wl);   // wave w owns output channels 16w .. 16w+15
    const float bz = bias[wave * 16 + l15];
    const float a = slope[0];
    // LayerNorm affine of this thread's fixed float4 slots, resident for all frames of the persistent loop
    constexpr int N = NF * C, N4 = N / 4;
    constexpr int NSLOT = (N4 + 255) / 256;      // 7
    float4 pw[NSLOT], pb[NSLOT];
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        const int i = min(tid + 256 * k, N4 - 1);
        pw[k] = *reinterpret_cast<const float4*>(&lnw[i * 4]);
        pb[k] = *reinterpret_cast<const float4*>(&lnb[i * 4]);
    }

    frame_zero_pad(ahi, alo, tid);
    float4 stg[FR_NLD];
    if ((int)blockIdx.x < nframes) frame_load_heads(merged + (long)blockIdx.x * N, tid, stg);
    for (int fidx = blockIdx.x; fidx < nframes; fidx += gridDim.x) {     // grid-stride over frames (b*T + t)
        const int b = fidx / T;
        const long fr = (long)fidx * N;
        frame_store(ahi, alo, tid, stg);
        __syncthreads();                      // image complete; also orders the previous frame's reads of `ys`
        if (fidx + (int)gridDim.x < nframes) frame_load_heads(merged + (long)(fidx + gridDim.x) * N, tid, stg);

        // residual rows of this frame: loads in flight during the MFMA + statistics phases
        float4 rv[NSLOT];
#pragma unroll
        for (int k = 0; k < NSLOT; ++k)
            rv[k] = *reinterpret_cast<const float4*>(&y2[fr + (long)min(tid + 256 * k, N4 - 1) * 4]);

#pragma unroll 1
        for (int m = 0; m < FR_RP / 16; ++m) {
            const f32x4 acc = mma_tile<FR_RP, 2>(ahi, alo, m, g4, l15, wh, wl, bz);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m * 16 + g4 * 4 + r;
                if (row < NF) ys[row * YP + wave * 16 + l15] = prelu_f(acc[r], a);
            }
        }
        __syncthreads();

        // joint LayerNorm over all 97*64 values of the frame (flat index f*64 + c), float4 granules
        float4 v[NSLOT];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const int i = min(tid + 256 * k, N4 - 1);
            v[k] = *reinterpret_cast<const float4*>(&ys[(i >> 4) * YP + (i & 15) * 4]);
            if (tid + 256 * k < N4) s += v[k].x + v[k].y + v[k].z + v[k].w;
        }
        const float mean = block_sum_256(s, red) * (1.0f / N);
        float vs = 0.f;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            if (tid + 256 * k < N4) vs += dx * dx + dy * dy + dz * dz + dw * dw;
        }
        const float rstd = rsqrtf(block_sum_256(vs, red) * (1.0f / N) + LN_EPS);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            const int i = tid + 256 * k;
            if (i < N4) {
                float4 o;
                o.x = rv[k].x + (v[k].x - mean) * rstd * pw[k].x + pb[k].x;
                o.y = rv[k].y + (v[k].y - mean) * rstd * pw[k].y + pb[k].y;
                o.z = rv[k].z + (v[k].z - mean) * rstd * pw[k].z + pb[k].z;
                o.w = rv[k].w + (v[k].w - mean) * rstd * pw[k].w + pb[k].w;
                if (gain) {
                    const float4 gv = *reinterpret_cast<const float4*>(&gain[(long)b * N + i * 4]);
                    o.x *= gv.x; o.y *= gv.y; o.z *= gv.z; o.w *= gv.w;
                }
                *reinterpret_cast<float4*>(&out[fr + i * 4]) = o;
            }
        }
    }
}

}  // namespace lh

extern "C" int lh_linear_res(const float* h, const void* w_pk, const float* bias, const float* res, float* out,
                             int rows, int K, lh_stream_t stream) {
    using namespace lh;
    if (!h || !w_pk || !bias || !res || !out || rows <= 0) return LH_ERR_ARG;
    const int ntiles = (rows + 63) / 64;
    const int grid = ntiles < 768 ? ntiles : 768;
    if (K == 128)
        hipLaunchKernelGGL((k_linear_res<128>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, (const _Float16*)w_pk,
                           bias, res, out, rows);
    else if (K == 64)
        hipLaunchKernelGGL((k_linear_res<64>), dim3(grid), dim3(256), 0, (hipStream_t)stream, h, (const _Float16*)w_pk,
                           bias, res, out, rows);
    else
        return LH_ERR_UNSUPPORTED;
    return check_launch();
}

extern "C" int lh_qkv_proj_ln(const float* y, const void* w_pk, const float* bias, const float* slopes,
                              const float* lnq_w, const float* lnq_b, const float* lnk_w, const float* lnk_b,
                              const float* lnv_w, const float* lnv_b, float* q, float* kx, float* vx, int B, int T,
                              lh_stream_t stream) {
    using namespace lh;
    if (!y || !w_pk || !bias || !slopes || !lnq_w || !lnq_b || !lnk_w || !lnk_b || !lnv_w || !lnv_b || !q || !kx ||
        !vx || B <= 0 || T <= 0)
        return LH_ERR_ARG;
    const int nframes = B * T;
    hipLaunchKernelGGL(k_qkv_proj_ln, dim3(nframes < 512 ? nframes : 512), dim3(256), 0, (hipStream_t)stream, y,
                       (const _Float16*)w_pk, bias, slopes, lnq_w, lnq_b, lnk_w, lnk_b, lnv_w, lnv_b, q, kx, vx, T,
                       nframes);
    return check_launch();
}

extern "C" int lh_proj_ln_res(const float* merged, const void* w_pk, const float* bias, const float* slope,
                              const float* ln_w, const float* ln_b, const float* y2, const float* gain, float* out,
                              int B, int T, lh_stream_t stream) {
    using namespace lh;
    if (!merged || !w_pk || !bias || !slope || !ln_w || !ln_b || !y2 || !out || B <= 0 || T <= 0) return LH_ERR_ARG;
    const int nframes = B * T;
    hipLaunchKernelGGL(k_proj_ln_res, dim3(nframes < 512 ? nframes : 512), dim3(256), 0, (hipStream_t)stream, merged,
                       (const _Float16*)w_pk, bias, slope, ln_w, ln_b, y2, gain, out, T, nframes);
    return check_launch();
}
