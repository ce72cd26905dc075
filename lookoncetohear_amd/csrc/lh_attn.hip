// Local (50-slot) causal attention with fused head merge, SURVEY.md §8a rows a15-a18.
//
// The reference materialises a 50x sliding-window copy of K and V (`_causal_unfold_chunk`,
// tfgridnet_causal.py:429-454: 67 % of its CPU time).  Here a workgroup owns TQ = 16*MQ consecutive query frames of
// one (batch, head): the TQ+49 history-extended K/V rows they can see are read ONCE from the ring-extended buffers.
//
// Q, K and V arrive as split-precision fp16 pairs (v = hi + lo, lo un-rescaled; written by k_qkv_proj_ln in exactly the
// order the v_mfma_f32_16x16x32_f16 operands want, lh_common.h), so both contractions run as three fp16 MFMAs per tile
// (hi*hi + hi*lo + lo*hi into one accumulator, ~22 mantissa bits) with no conversion work in this kernel:
//   * scores  S = Q K^T: banded tile products (query tile mq x key tiles mq..mq+4), the 19 feature k-steps split
//     over the 4 waves, partial sums reduced through LDS; A and B fragments are 32-byte row segments straight from
//     global memory (two-deep register ring);
//   * softmax over exactly the 50 in-window slots (zero history rows take part, no mask — reference behaviour),
//     P written to LDS as fp16 hi/lo rows;
//   * O = P V: per 32-key step a lane loads the [hi 4 | lo 4] quads of 8 key rows for its 4 columns (256-byte row
//     segments per 16 lanes) and regroups them lane-locally into the four column tiles' B operands; the three key
//     steps form a register ring that is refilled for the wave's next column group right after use.
// The output keeps the head-major order [B][T][head][97][16] (full-line stores) — the projection kernel reads it in
// that order.  Workgroups of one (batch, head) are placed on one XCD (blockIdx % 8) so neighbouring tiles re-read
// the shared 49 rows from that XCD's L2 instead of HBM.  kx / vx carry KV_PAD zero rows behind row T+48: tiles
// read (never need) up to 47 rows past their last key and 0 * finite = 0.
#include "lh_common.h"

namespace lh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
struct Frag { f16x8 h, l; };               // 8 k-values of one operand row: hi and lo halves
struct VQuad { f16x4 h, l; };              // one key row, 4 columns

constexpr int AT_KSTEPS = (DQKP + 31) / 32;   // 19 feature k-steps of 32
constexpr int AT_PKS = 3;                     // 32-key steps in P.V (96 >= 32 + 49)
constexpr int AT_PP = AT_PKS * 32 + 8;        // P row stride in halves (208 bytes: conflict-free ds_read_b128)
constexpr int AT_CG = (DV + 63) / 64;         // 25 column groups of 64 V columns

#if defined(LH_PROBE_TRACE)               // timing probe build only (scripts/probe_trace.py): workgroup 803, wave 0
__device__ unsigned long long lh_attn_trace_buf[16];
#define AT_STAMP(k) do { if (blockIdx.x == 803 && blockIdx.y == 0 && tid == 0) lh_attn_trace_buf[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AT_STAMP(k) do { } while (0)
#endif
template <int MQ, int RING, int TQ = 16 * MQ>
__global__ void __launch_bounds__(256, 2) k_local_attn(const _Float16* __restrict__ q, const _Float16* __restrict__ kx,
                                                       const _Float16* __restrict__ vx, float* __restrict__ merged,
                                                       int BH, int T, int ntt, int tw0, int tend) {
    // TQ query frames per workgroup in MQ MFMA row tiles (TQ = 16 MQ, or 40 in 3 tiles: 625 frames are then 16 tiles
    // per (batch, head), 2048 workgroups at batch 32 = exactly four rounds of the 512 resident workgroups)
    constexpr int NKEYS = TQ + HIST;           // key rows they can see (65 / 81 / 89)
    constexpr int NKT = (NKEYS + 15) / 16;     // key tiles in the score phase (5 / 6)
    constexpr int NK = NKT * 16;
    constexpr int ND = 5;                      // key tiles per query tile: nt = mq .. mq+4 (as far as they exist)
    constexpr int NB = ND * 16;                // band width stored per query row
    static_assert(TQ <= 16 * MQ && TQ > 16 * (MQ - 1) && NKEYS <= AT_PKS * 32 && NK - 1 <= HIST + KV_PAD, "tile geometry");
    __shared__ float sp[4][TQ][NB];            // per-wave partial scores, band-relative columns (d, l15)
    __shared__ __attribute__((aligned(16))) _Float16 ph[16 * MQ][AT_PP];
    __shared__ __attribute__((aligned(16))) _Float16 pl[16 * MQ][AT_PP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid), g4 = lane >> 4, l15 = lane & 15;
    // XCD-aware placement: the tiles of (batch, head) bh all run on XCD bh % 8
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int bh = (kk / ntt) * 8 + xcd;
    const int t0 = tw0 + (kk % ntt) * TQ;        // time window [tw0, tend) of the T frames (lh_local_attn_win; whole clip: 0, T)
    if (bh >= BH) return;
    const long TKP = T + HIST + KV_PAD;
    const _Float16* qb = q + (long)bh * T * LDQKH;
    const _Float16* kb = kx + ((long)bh * TKP + t0) * LDQKH;
    const _Float16* vb = vx + ((long)bh * TKP + t0) * LDVH;

    AT_STAMP(0);
    // ---- scores: S[i][n] = <Q[t0+i], Kx[t0+n]>, feature k-steps s = wave, wave+4, ... of 19
    {
        f32x4 am[MQ][ND];
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq)
#pragma unroll
            for (int d = 0; d < ND; ++d) am[mq][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        int qoff[MQ], koff[NKT];
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq) qoff[mq] = min(t0 + mq * 16 + l15, tend - 1) * LDQKH + g4 * 16;
#pragma unroll
        for (int nt = 0; nt < NKT; ++nt) koff[nt] = (nt * 16 + l15) * LDQKH + g4 * 16;
        constexpr int NIT = (AT_KSTEPS + 3) / 4;       // 5 (wave 3 runs 4)
        Frag fq[2][MQ], fk[2][NKT];
        auto fetch = [&](int it, Frag (&dq)[MQ], Frag (&dk)[NKT]) {
            const int s = min(wave + 4 * it, AT_KSTEPS - 1);
#pragma unroll
            for (int mq = 0; mq < MQ; ++mq) {
                const _Float16* p = qb + qoff[mq] + s * 64;
                dq[mq].h = *reinterpret_cast<const f16x8*>(p);
                dq[mq].l = *reinterpret_cast<const f16x8*>(p + 8);
            }
#pragma unroll
            for (int nt = 0; nt < NKT; ++nt) {
                const _Float16* p = kb + koff[nt] + s * 64;
                dk[nt].h = *reinterpret_cast<const f16x8*>(p);
                dk[nt].l = *reinterpret_cast<const f16x8*>(p + 8);
            }
        };
        auto mma = [&](const Frag (&sq)[MQ], const Frag (&sk)[NKT]) {
#pragma unroll
            for (int mq = 0; mq < MQ; ++mq)
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    if (mq + d >= NKT) continue;       // compile-time: the last tile of a 40-frame workgroup sees 4 key tiles
                    const Frag& kf = sk[mq + d];
                    am[mq][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sq[mq].h, kf.h, am[mq][d], 0, 0, 0);
                    am[mq][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sq[mq].h, kf.l, am[mq][d], 0, 0, 0);
                    am[mq][d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sq[mq].l, kf.h, am[mq][d], 0, 0, 0);
                }
        };
        fetch(0, fq[0], fk[0]);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (it + 1 < NIT) fetch(it + 1, fq[(it + 1) & 1], fk[(it + 1) & 1]);
            if (wave + 4 * it < AT_KSTEPS) mma(fq[it & 1], fk[it & 1]);       // wave-uniform
        }
#pragma unroll
        for (int mq = 0; mq < MQ; ++mq)
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mq * 16 + g4 * 4 + r;
                    if (TQ == 16 * MQ || row < TQ) sp[wave][row][d * 16 + l15] = am[mq][d][r];
                }
    }
    AT_STAMP(1);
    __syncthreads();
    AT_STAMP(2);

    // ---- softmax over the 50 slots n = i .. i+49 of query i; 16 threads per query, P as fp16 hi/lo rows
    {
        const float scale = 1.0f / sqrtf((float)DQK);
        constexpr int NU = AT_PKS * 2;            // 6 column slots of 16 per thread (96 key columns)
#pragma unroll
        for (int pass = 0; pass < MQ; ++pass) {
            const int ir = pass * 16 + (tid >> 4), sub = tid & 15;
            const int i = min(ir, TQ - 1);         // rows past TQ (40-frame tiles) recompute row TQ-1 and store nothing
            float sv[NU];
            float mx = -3.0e38f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int n = sub + 16 * u, j = n - i;
                const int c = n - pass * 16;           // band-relative column; in [0, 65) whenever j is in the window
                float s = -3.0e38f;
                if (j >= 0 && j < WIN) s = (sp[0][i][c] + sp[1][i][c] + sp[2][i][c] + sp[3][i][c]) * scale;
                sv[u] = s;
                mx = fmaxf(mx, s);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.f;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int j = sub + 16 * u - i;
                sv[u] = (j >= 0 && j < WIN) ? __expf(sv[u] - mx) : 0.f;
                sum += sv[u];
            }
            sum = group16_sum(sum);
            const float inv = 1.0f / sum;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float p = sv[u] * inv;           // exactly 0 outside the band
                _Float16 h, l;
                split_hl(p, h, l);
                if (TQ == 16 * MQ || ir < TQ) {
                    ph[ir][sub + 16 * u] = h;
                    pl[ir][sub + 16 * u] = l;
                }
            }
        }
    }
    AT_STAMP(3);
    __syncthreads();
    AT_STAMP(4);

    // ---- O = P . Vx ; each wave takes column groups of 64 (4 MFMA column tiles interleaved so a lane loads one
    //      [hi 4 | lo 4] quad of a V row per key); the head merge is fused into the store
    {
        const int b = bh / NH, hd = bh % NH;
        // gridDim.y workgroups share one (batch, head, tile): each recomputes the (cheap) scores and takes every
        // gridDim.y-th round of column groups — a chunk of ONE frame would otherwise keep 4 of the 256 CUs busy
        const int part = blockIdx.y, nsplit = gridDim.y;
        // key row of (step ks, slot j) for this lane: 32 ks + 8 g4 + j; rows >= NKEYS are never needed (P = 0 there)
        auto qcol = [&](int cg) { return min(min(cg, AT_CG - 1) * 16 + l15, DV / 4 - 1) * 8; };   // quad offset in halves
        // Register ring over the flattened (column group, key step) sequence of this wave: step s uses slot s % RING
        // and, before its MFMAs, refills the slot of step s-1 with the rows of step s+RING-1.
        VQuad ring[RING][8];
        // Every load is UNCONDITIONAL (round 5): a lane whose row is >= NKEYS (last key step only: P is exactly 0 there, and
        // 0 * finite = 0) reads the row the lanes one quarter below load in the SAME instruction (same cache lines, no extra
        // traffic) instead of skipping the load.  Behind the per-lane `row < NKEYS` branches of rounds 1-4 hipcc lost count
        // of the loads in flight and drained them all (s_waitcnt vmcnt(0) right behind the refill, once per column group:
        // the whole L2 latency exposed); in straight-line code the waits are exact and the refill stays in flight.
        auto fill = [&](int slot, int ks, int qc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int r = 32 * ks + 8 * g4 + j;
                if (32 * ks + 31 >= NKEYS) {                   // (compile-time per unrolled step: the last key step only)
                    r = r >= NKEYS ? r - 8 : r;
                    r = r >= NKEYS ? NKEYS - 1 : r;
                }
                const _Float16* p = vb + r * LDVH + qc;
                VQuad v;
                v.h = *reinterpret_cast<const f16x4*>(p);
                v.l = *reinterpret_cast<const f16x4*>(p + 4);
                ring[slot][j] = v;
            }
        };
#pragma unroll
        for (int s = 0; s < RING - 1; ++s) fill(s % RING, s % AT_PKS, qcol(wave + 4 * (part + nsplit * (s / AT_PKS))));
        f32x4 am[MQ][4];
        // fully unrolled (7 column groups for wave 0, 6 for the others): exact vmcnt waits instead of a drain of the
        // ring at every loop back-edge
        constexpr int NSTEP = ((AT_CG + 3) / 4) * AT_PKS;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int ks = s % AT_PKS, cg = wave + 4 * (part + nsplit * (s / AT_PKS));
            if (cg >= AT_CG) break;                    // wave-uniform
            if (ks == 0) {
#pragma unroll
                for (int mq = 0; mq < MQ; ++mq)
#pragma unroll
                    for (int c = 0; c < 4; ++c) am[mq][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            {
                const int sn = s + RING - 1, cgn = wave + 4 * (part + nsplit * (sn / AT_PKS));
                fill(sn % RING, sn % AT_PKS, qcol(cgn));      // (no `cgn < AT_CG` branch: qcol clamps, the wave's last refill is a re-read)
                // the refill goes out HERE: without the fence hipcc sinks the loads below this step's MFMAs, next to their
                // first use (it saves the ring's registers that way) and every step waits out a full V-row load
                __builtin_amdgcn_sched_barrier(0);
            }
            // regroup: column tile c takes element c of the 8 key rows -> one f16x8 B operand (hi and lo)
            f16x8 bh8[4], bl8[4];
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    bh8[c][j] = ring[s % RING][j].h[c];
                    bl8[c][j] = ring[s % RING][j].l[c];
                }
#pragma unroll
            for (int mq = 0; mq < MQ; ++mq) {
                const f16x8 pa = *reinterpret_cast<const f16x8*>(&ph[mq * 16 + l15][ks * 32 + g4 * 8]);
                const f16x8 pb = *reinterpret_cast<const f16x8*>(&pl[mq * 16 + l15][ks * 32 + g4 * 8]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    am[mq][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, bh8[c], am[mq][c], 0, 0, 0);
                    am[mq][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, bl8[c], am[mq][c], 0, 0, 0);
                    am[mq][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pb, bh8[c], am[mq][c], 0, 0, 0);
                }
            }
            const int col = cg * 64 + l15 * 4;
            if (ks == AT_PKS - 1 && col < DV) {
#pragma unroll
                for (int mq = 0; mq < MQ; ++mq)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int t = t0 + mq * 16 + g4 * 4 + r;
                        if (t < tend && (TQ == 16 * MQ || mq * 16 + g4 * 4 + r < TQ))      // head-major slab [b][t][hd][f][v]: one head's frame is 6208 contiguous bytes
                            *reinterpret_cast<float4*>(&merged[(((long)b * T + t) * NH + hd) * DV + col]) =
                                make_float4(am[mq][0][r], am[mq][1][r], am[mq][2][r], am[mq][3][r]);
                    }
            }
        }
    }
    AT_STAMP(5);
}

// ------------------------------------------------------------------------------------------------------
// Streaming state <-> ring rows: the reference carries K_buf [4B][49][582] / V_buf [4B][49][1552] as fp32
// (tfgridnet_causal.py:553-562); the history rows of kx / vx hold the same numbers as split fp16 pairs.
//   pack:   rows 0..48 of kx / vx   <- K_buf / V_buf
//   unpack: K_buf / V_buf           <- rows T..T+48 of kx / vx  (hi + lo: the value the attention kernel used)
// ------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ring_pack(const float* __restrict__ kbuf, const float* __restrict__ vbuf,
                                                   _Float16* __restrict__ kx, _Float16* __restrict__ vx, long tkp, int BH) {
    const long nk = (long)BH * HIST * QKB, nv = (long)BH * HIST * (DV / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nk + nv; i += (long)gridDim.x * 256) {
        if (i < nk) {                                   // one 8-feature block of a K row
            const long row = i / QKB;
            const int blk = (int)(i % QKB);
            const long bh = row / HIST, r = row % HIST;
            f16x8 h8, l8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = blk * 8 + e;
                const float v = f < DQK ? kbuf[row * DQK + f] : 0.f;
                _Float16 h, l;
                split_hl(v, h, l);
                h8[e] = h;
                l8[e] = l;
            }
            _Float16* d = kx + (bh * tkp + r) * LDQKH + blk * 16;
            *reinterpret_cast<f16x8*>(d) = h8;
            *reinterpret_cast<f16x8*>(d + 8) = l8;
        } else {                                        // one 4-column quad of a V row
            const long k = i - nk;
            const long row = k / (DV / 4);
            const int qd = (int)(k % (DV / 4));
            const long bh = row / HIST, r = row % HIST;
            const float4 v4 = *reinterpret_cast<const float4*>(&vbuf[row * DV + qd * 4]);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 h, l;
                split_hl(v[e], h, l);
                o[e] = h;
                o[4 + e] = l;
            }
            *reinterpret_cast<f16x8*>(vx + (bh * tkp + r) * LDVH + qd * 8) = o;
        }
    }
}

__global__ void __launch_bounds__(256) k_ring_unpack(const _Float16* __restrict__ kx, const _Float16* __restrict__ vx,
                                                     float* __restrict__ kbuf, float* __restrict__ vbuf, long tkp, int T,
                                                     int BH) {
    const long nk = (long)BH * HIST * QKB, nv = (long)BH * HIST * (DV / 4);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nk + nv; i += (long)gridDim.x * 256) {
        if (i < nk) {
            const long row = i / QKB;
            const int blk = (int)(i % QKB);
            const long bh = row / HIST, r = row % HIST;
            const _Float16* s = kx + (bh * tkp + T + r) * LDQKH + blk * 16;
            const f16x8 h8 = *reinterpret_cast<const f16x8*>(s), l8 = *reinterpret_cast<const f16x8*>(s + 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = blk * 8 + e;
                if (f < DQK) kbuf[row * DQK + f] = (float)h8[e] + (float)l8[e];
            }
        } else {
            const long k = i - nk;
            const long row = k / (DV / 4);
            const int qd = (int)(k % (DV / 4));
            const long bh = row / HIST, r = row % HIST;
            const f16x8 o = *reinterpret_cast<const f16x8*>(vx + (bh * tkp + T + r) * LDVH + qd * 8);
            *reinterpret_cast<float4*>(&vbuf[row * DV + qd * 4]) =
                make_float4((float)o[0] + (float)o[4], (float)o[1] + (float)o[5], (float)o[2] + (float)o[6],
                            (float)o[3] + (float)o[7]);
        }
    }
}

}  // namespace lh
#if defined(LH_PROBE_TRACE)
extern "C" int lh_probe_attn_trace_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::lh_attn_trace_buf), sizeof(lh::lh_attn_trace_buf)) == hipSuccess ? 0 : 1;
}
#endif
namespace lh {
static int g_attn_mq = 0;              // lh_set_tuning key 4: 0 = automatic, 1 / 2 = query tiles of 16 frames per workgroup,
                                       // 3 = 40 frames in three tiles
int attn_set_mq(int v) {
    if (v < 0 || v > 3) return LH_ERR_ARG;
    g_attn_mq = v;
    return LH_OK;
}

}  // namespace lh

extern "C" int lh_local_attn_win(const void* q, const void* kx, const void* vx, float* merged, int B, int T, int t0, int Tc,
                                 lh_stream_t stream) {
    using namespace lh;
    if (!q || !kx || !vx || !merged || B <= 0 || T <= 0 || t0 < 0 || Tc <= 0 || t0 + Tc > T) return LH_ERR_ARG;
    const int Tfull = T;
    // tile shape from the WHOLE clip (a window then cuts the time axis into the same tiles as the whole-clip launch when t0
    // is a multiple of the tile length: bit-identical outputs); tile count and bounds from the window
    const int Tshape = T;
    T = Tc;
    const int BH = B * NH;
    const int bh8 = (BH + 7) / 8 * 8;
    // Query frames per workgroup: one 16-frame tile when the clip is that short (streaming: T = 1); two tiles share
    // every K / V row they load (25 instead of 44 KB of L2 traffic per query; 0.36 against 0.43 ms at B = 32); 40 frames
    // in three tiles once the launch has more workgroups than the 512 resident ones (12 % fewer K / V rows per query,
    // and at B = 32, T = 625 exactly four rounds of workgroups instead of five: 0.343 against 0.356 ms) — below that a
    // launch is latency-bound and the shorter workgroup wins.  The V ring is two-deep: a three-deep one (228 registers
    // since the single-accumulator split) measured 3 % slower.
    int mq = g_attn_mq;
    if (Tshape <= 16) mq = 1;
    else if (mq == 0) mq = (long)bh8 * ((Tshape + 31) / 32) > 512 ? 3 : 2;
    const int tq = mq == 3 ? 40 : 16 * mq;
    const int ntt = (T + tq - 1) / tq;
    // latency-bound launches (a handful of workgroups): split the V columns of a tile over 7 workgroups
    const dim3 grid(bh8 * ntt, bh8 * ntt <= 64 ? 7 : 1);
    const _Float16 *qh = (const _Float16*)q, *kh = (const _Float16*)kx, *vh = (const _Float16*)vx;
    if (mq == 1)
        hipLaunchKernelGGL((k_local_attn<1, 3>), grid, dim3(256), 0, (hipStream_t)stream, qh, kh, vh, merged, BH, Tfull, ntt, t0, t0 + Tc);
    else if (mq == 3)
        hipLaunchKernelGGL((k_local_attn<3, 2, 40>), grid, dim3(256), 0, (hipStream_t)stream, qh, kh, vh, merged, BH, Tfull, ntt, t0, t0 + Tc);
    else
        hipLaunchKernelGGL((k_local_attn<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, qh, kh, vh, merged, BH, Tfull, ntt, t0, t0 + Tc);
    return check_launch();
}

extern "C" int lh_local_attn(const void* q, const void* kx, const void* vx, float* merged, int B, int T,
                             lh_stream_t stream) {
    return lh_local_attn_win(q, kx, vx, merged, B, T, 0, T, stream);
}

extern "C" int lh_ring_pack(const float* k_buf, const float* v_buf, void* kx, void* vx, int B, int T,
                            lh_stream_t stream) {
    using namespace lh;
    if (!k_buf || !v_buf || !kx || !vx || B <= 0 || T <= 0) return LH_ERR_ARG;
    const int BH = B * NH;
    const long n = (long)BH * HIST * (QKB + DV / 4);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_ring_pack, dim3(grid), dim3(256), 0, (hipStream_t)stream, k_buf, v_buf, (_Float16*)kx,
                       (_Float16*)vx, (long)T + HIST + KV_PAD, BH);
    return check_launch();
}

namespace lh {
__global__ void k_ring_advance(int* pos, int modulo) {
    const int p = *pos + 1;
    *pos = p >= modulo ? p - modulo : p;
}
}  // namespace lh

extern "C" int lh_ring_advance(int* ring_pos, int modulo, lh_stream_t stream) {
    if (!ring_pos || modulo < 1 || modulo > (1 << 30)) return LH_ERR_ARG;
    hipLaunchKernelGGL(lh::k_ring_advance, dim3(1), dim3(1), 0, (hipStream_t)stream, ring_pos, modulo);
    return lh::check_launch();
}

extern "C" int lh_ring_unpack(const void* kx, const void* vx, float* k_buf, float* v_buf, int B, int T,
                              lh_stream_t stream) {
    using namespace lh;
    if (!k_buf || !v_buf || !kx || !vx || B <= 0 || T <= 0) return LH_ERR_ARG;
    const int BH = B * NH;
    const long n = (long)BH * HIST * (QKB + DV / 4);
    const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_ring_unpack, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const _Float16*)kx,
                       (const _Float16*)vx, k_buf, v_buf, (long)T + HIST + KV_PAD, T, BH);
    return check_launch();
}
