// Local (50-slot) causal attention with fused head merge, SURVEY.md §8a rows a15-a18.
//
// The reference materialises a 50x sliding-window copy of K and V (`_causal_unfold_chunk`,
// tfgridnet_causal.py:429-454: 67 % of its CPU time).  Here a workgroup owns 16 consecutive query frames of
// one (batch, head): the 65 history-extended K/V rows they can see are read ONCE from the ring-extended
// buffers, scores are a banded 16 x 80 fp32-MFMA product split over the feature axis across the 4 waves,
// softmax runs over exactly the 50 in-window slots (zero history rows take part, no mask — reference
// behaviour), and P.V streams V rows as 256-byte float4 row segments straight into MFMA B operands.  The output
// keeps the head-major order [B][T][head][97][16] (full-line stores; interleaving the heads at 64-byte granularity
// made every cache line a partial write of two workgroups) — the projection kernel reads it in that order.
// Workgroups of one (batch, head) are placed on one XCD (blockIdx % 8) so neighbouring tiles re-read the
// shared 49 rows from that XCD's L2 instead of HBM.
#include "lh_common.h"

namespace lh {

constexpr int AT_TQ = 16;                  // query frames per workgroup
constexpr int AT_NKT = 5;                  // key tiles of 16 -> 80 >= 16 + 49 rows
constexpr int AT_NK = AT_NKT * 16;         // 80
constexpr int AT_KS = 17;                  // k-steps of 4 keys in P.V (68 >= 65)
constexpr int AT_PP = AT_KS * 4 + 4;       // P row stride (72)
constexpr int AT_F4 = LDQK / 4;            // 146 float4 per q/k row
constexpr int AT_CG = (DV + 63) / 64;      // 25 column groups of 64 V columns

__global__ void __launch_bounds__(256) k_local_attn(const float* __restrict__ q, const float* __restrict__ kx,
                                                    const float* __restrict__ vx, float* __restrict__ merged,
                                                    int BH, int T, int ntt) {
    __shared__ float sp[4][AT_TQ][AT_NK];
    __shared__ float pm[AT_TQ][AT_PP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = lane >> 4, l15 = lane & 15;
    // XCD-aware placement: the tiles of (batch, head) bh all run on XCD bh % 8
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int bh = (kk / ntt) * 8 + xcd;
    const int t0 = (kk % ntt) * AT_TQ;
    if (bh >= BH) return;
    const int TK = T + HIST;
    const float* qb = q + (long)bh * T * LDQK;
    const float* kb = kx + (long)bh * TK * LDQK;
    const float* vb = vx + (long)bh * TK * DV;

    // ---- scores: S[i][n] = <Q[t0+i], Kx[t0+n]>, feature axis split over waves and 16-lane groups
    {
        f32x4 acc[AT_NKT];
#pragma unroll
        for (int nt = 0; nt < AT_NKT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* qrow = qb + (long)min(t0 + l15, T - 1) * LDQK;
        const float* krow[AT_NKT];
#pragma unroll
        for (int nt = 0; nt < AT_NKT; ++nt) krow[nt] = kb + (long)min(t0 + nt * 16 + l15, TK - 1) * LDQK;
        // two-deep register ring over the feature iterations: the 6 row segments of iteration it+1 are in flight
        // while the 20 MFMAs of iteration it run (the compiler otherwise waits for every load right before its use)
        constexpr int NIT = (AT_F4 + 15) / 16;
        float4 ring[2][AT_NKT + 1];
        auto fetch = [&](int it, float4 (&dst)[AT_NKT + 1]) {
            const int f4 = it * 16 + wave * 4 + g4;
            const int off = f4 < AT_F4 ? f4 * 4 : 0;
            dst[AT_NKT] = *reinterpret_cast<const float4*>(qrow + off);
#pragma unroll
            for (int nt = 0; nt < AT_NKT; ++nt) dst[nt] = *reinterpret_cast<const float4*>(krow[nt] + off);
        };
        auto mma = [&](int it, const float4 (&src)[AT_NKT + 1]) {
            const bool ok = it * 16 + wave * 4 + g4 < AT_F4;
            const float4 a4 = src[AT_NKT];
            const float av[4] = {ok ? a4.x : 0.f, ok ? a4.y : 0.f, ok ? a4.z : 0.f, ok ? a4.w : 0.f};
#pragma unroll
            for (int nt = 0; nt < AT_NKT; ++nt) {
                const float bv[4] = {src[nt].x, src[nt].y, src[nt].z, src[nt].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[nt], 0, 0, 0);
            }
        };
        static_assert(NIT % 2 == 0, "ring parity");
        fetch(0, ring[0]);
#pragma unroll 1
        for (int it = 0; it < NIT; it += 2) {
            fetch(it + 1, ring[1]);
            mma(it, ring[0]);
            fetch(min(it + 2, NIT - 1), ring[0]);
            mma(it + 1, ring[1]);
        }
#pragma unroll
        for (int nt = 0; nt < AT_NKT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) sp[wave][g4 * 4 + r][nt * 16 + l15] = acc[nt][r];
    }
    __syncthreads();

    // ---- softmax over the 50 slots n = i .. i+49 of query i; 16 threads per query
    {
        const int i = tid >> 4, sub = tid & 15;
        const float scale = 1.0f / sqrtf((float)DQK);
        float sv[4];
        float mx = -3.0e38f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = sub + 16 * u;
            float s = -3.0e38f;
            if (j < WIN) {
                const int n = i + j;
                s = (sp[0][i][n] + sp[1][i][n] + sp[2][i][n] + sp[3][i][n]) * scale;
            }
            sv[u] = s;
            mx = fmaxf(mx, s);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = sub + 16 * u;
            sv[u] = (j < WIN) ? __expf(sv[u] - mx) : 0.f;
            sum += sv[u];
        }
        sum = group16_sum(sum);
        const float inv = 1.0f / sum;
        for (int n = sub; n < AT_PP; n += 16) pm[i][n] = 0.f;      // outside the band
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = sub + 16 * u;
            if (j < WIN) pm[i][i + j] = sv[u] * inv;
        }
    }
    __syncthreads();

    // ---- O = P . Vx ; each wave takes column groups of 64 (4 MFMA column tiles interleaved so a lane loads
    //      one float4 of a V row per k-step); the head merge is fused into the store
    {
        float pa[AT_KS];
#pragma unroll
        for (int ks = 0; ks < AT_KS; ++ks) pa[ks] = pm[l15][ks * 4 + g4];
        const int b = bh / NH, hd = bh % NH;
        // V-row ring: slot ks is refilled with the next column group's row segment right after its 4 MFMAs, so
        // every load has the other 16 k-steps (64 MFMAs) to land
        int vrow[AT_KS];
#pragma unroll
        for (int ks = 0; ks < AT_KS; ++ks) vrow[ks] = min(t0 + ks * 4 + g4, TK - 1);
        auto vcol = [&](int cg) { const int c = min(cg, AT_CG - 1) * 64 + l15 * 4; return c < DV ? c : 0; };
        float4 vr[AT_KS];
        {
            const int lc = vcol(wave);
#pragma unroll
            for (int ks = 0; ks < AT_KS; ++ks) vr[ks] = *reinterpret_cast<const float4*>(vb + (long)vrow[ks] * DV + lc);
        }
        // fully unrolled (7 column groups for wave 0, 6 for the others): exact vmcnt waits instead of a drain of the
        // ring at every loop back-edge
#pragma unroll
        for (int ci = 0; ci < (AT_CG + 3) / 4; ++ci) {
            const int cg = wave + 4 * ci;
            if (cg >= AT_CG) break;                    // wave-uniform
            const int col = cg * 64 + l15 * 4;
            const bool colok = col < DV;
            const int ncol = vcol(cg + 4);
            f32x4 acc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < AT_KS; ++ks) {
                const float4 v4 = vr[ks];
                vr[ks] = *reinterpret_cast<const float4*>(vb + (long)vrow[ks] * DV + ncol);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ks], v4.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ks], v4.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ks], v4.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[ks], v4.w, acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);      // keep the refill next to its slot's MFMAs (the scheduler sinks it)
            }
            if (colok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int t = t0 + g4 * 4 + r;
                    if (t < T)      // head-major slab [b][t][hd][f][v]: one head's frame is 6208 contiguous bytes
                        *reinterpret_cast<float4*>(&merged[(((long)b * T + t) * NH + hd) * DV + col]) =
                            make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
                }
            }
        }
    }
}

}  // namespace lh

extern "C" int lh_local_attn(const float* q, const float* kx, const float* vx, float* merged, int B, int T,
                             lh_stream_t stream) {
    using namespace lh;
    if (!q || !kx || !vx || !merged || B <= 0 || T <= 0) return LH_ERR_ARG;
    const int BH = B * NH;
    const int ntt = (T + AT_TQ - 1) / AT_TQ;
    const int bh8 = (BH + 7) / 8 * 8;
    hipLaunchKernelGGL(k_local_attn, dim3(bh8 * ntt), dim3(256), 0, (hipStream_t)stream, q, kx, vx, merged, BH, T, ntt);
    return check_launch();
}
