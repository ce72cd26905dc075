// Software-pipelined fused recurrences, round 3 (reference tfgridnet_causal.py:505-538).
//
// k_intra_xp: LayerNorm + one direction of the intra BiLSTM + that direction's half of the output Linear + residual.
// Same tile, LDS images, weight images and HBM traffic as k_ln_lstm_lin<1> (lh_lstm.hip) — 16 sequences x 97 steps per
// 4-wave workgroup, two workgroups per CU, [W_ih | W_hh] resident in VGPRs as f16x3 B fragments — but the step is
// re-cut along its DEPENDENCES so that every matrix instruction has independent vector work to issue beside it:
//
//   phase H (on the chain):  gates = gx + h_{t-1} W_hh^T            24 MFMAs  ||  the step's row-wise work: LayerNorm + split
//                                                                               of x_{t+2}, finished rows of step t-2, loads
//   phase C (off the chain): lin(h_{t-1}) (6 MFMAs), gx' = LN(x_{t+1}) W_ih'^T (24 MFMAs, the NON-recurrent half of step
//                            t+1's gates, one step ahead)          30 MFMAs  ||  the cell update of step t (40 transcendentals)
//
// and the instruction order inside each phase is written down: a "zipper" (xp_zip) issues one MFMA, then its slice of a
// list of small vector operations, then a scheduling fence.  (sched_group_barrier patterns did not produce the interleave
// on this code — the MFMAs stayed in runs of 8..24 — and the fenced zipper only survives without the SLP vectoriser, which
// otherwise gathers the scalar fp32 operations of different slots into packed instructions at one place: build.py.)
// The measured issue model (profiles/r02a_ubench_issue_model.txt): a 16x16x32 MFMA holds the matrix pipe 19 cycles but
// the wave's issue only ~7.5; two plain vector instructions (or one transcendental) behind it are free.  Result
// (profiles/r03a_*): a wave-step 2886 -> 2088 cycles alone on the CU, 4212 -> ~2690 with a second workgroup — and the
// shader clock 1929 -> 1788 MHz at the same 1.38 kW package power: the call gains 6 %.  The path is power-bound
// (DESIGN.md section 5).
//
// Further instruction diet against k_ln_lstm_lin: the gate bias leaves the accumulator initialisation (16 v_mov per
// step) and enters the cell update as a factor, 1 + 2^(a + b) = fma(2^a, 2^b, 1) with 2^b held per lane; the cell state
// is carried pre-scaled by -2 log2 e so that tanh(c) needs no multiply.
#include <tuple>
#include <type_traits>
#include <utility>

#include "lh_common.h"

namespace lh {

typedef _Float16 xp_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xp_f16x4 __attribute__((ext_vector_type(4)));
constexpr int XP_AP = 144;      // fp16 elements per LDS row: [x 64 | h 64] + 16 pad (288 B: conflict-free ds_read_b128)
constexpr int XP_LSP = 68;      // fp32 projection rows: 64 + 4 pad
constexpr float XP_K2 = -2.0f * LOG2E;

// compile-time loop helpers xp_for / xp_sf: lh_common.h (shared with the embedder's recurrent kernel)
// slot i of NSLOT: mf(i), then operations [i*NOPS/NSLOT, (i+1)*NOPS/NSLOT) of the tuple `ops`, then a scheduling fence
template <int NSLOT, class MF, class Ops>
__device__ __forceinline__ void xp_zip(MF& mf, Ops& ops) {
    constexpr int NOPS = (int)std::tuple_size<Ops>::value;
    xp_for<NSLOT>([&](auto s_) {
        constexpr int i = decltype(s_)::value;
        mf(s_);
        constexpr int a = i * NOPS / NSLOT, b = (i + 1) * NOPS / NSLOT;
        xp_for<b - a>([&](auto k_) { std::get<a + decltype(k_)::value>(ops)(); });
#if !defined(XP_NO_FENCE)
        __builtin_amdgcn_sched_barrier(0);
#endif
    });
}

#if defined(XP_X2)             // timing / error probe: x half as hi*hi + hi*lo_W only (activations rounded to fp16)
constexpr int XP_XN = 8;
#else
constexpr int XP_XN = 12;      // MFMAs per k-step of the x half: 4 gates x (hi*hi, hi*lo, lo*hi)
#endif

#if defined(XP_PF1)             // A/B probe: one step ahead
constexpr int XP_PF = 1;
#else
constexpr int XP_PF = 2;       // k_inter_xp: global rows are fetched two steps (~1.8 us) ahead of their use
#endif

#if defined(XP_TRACE)          // timing probe build only: s_memtime stamps of one wave of two workgroups of k_intra_xp
__device__ unsigned long long xp_trace_buf[2 * 128 * 4];
__device__ unsigned long long xp_wg_log[4096 * 4];       // per workgroup: start tick, end tick, HW_ID, XCC_ID
__device__ unsigned long long xp_trace_inter[2 * 128 * 4];   // k_inter_xp, workgroup 3: wave 0 (projection role) | wave 4 (LayerNorm role)
#define XP_STAMP(k) do { if (tr_on) tr[(it & 127) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define XP_STAMP(k) do { } while (0)
#endif

// Energy-attribution probes (timing builds only, WRONG results on purpose; scripts/power_probe.py): every kernel of the
// path runs at the 1400 W package cap, so time is proportional to energy and removing one instruction class shows its share.
__device__ __forceinline__ float xp_exp2(float v) {
#if defined(XP_PROBE_NOTRANS)
    return __builtin_fmaf(v, 0.25f, 0.5f);
#else
    return __builtin_amdgcn_exp2f(v);
#endif
}
__device__ __forceinline__ float xp_rcp(float v) {
#if defined(XP_PROBE_NOTRANS)
    return __builtin_fmaf(v, -0.25f, 1.0f);
#else
    return __builtin_amdgcn_rcpf(v);
#endif
}
#if defined(XP_PROBE_HIHI)
constexpr int XP_NP = 1;       // probe: hi*hi products only (18 of 54 MFMAs)
#else
constexpr int XP_NP = 3;
#endif

__global__ void __launch_bounds__(256, 2) k_intra_xp(const float* __restrict__ x, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ b_sum, const _Float16* __restrict__ wlin_pk,
                                                     const float* __restrict__ blin, float* out, int nseq, int nstep,
                                                     int sdiv, int so, int si, int ps, int dir, int accumulate, int prio) {
    constexpr int NS = 16;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * NS * XP_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * NS * XP_AP];
    __shared__ __attribute__((aligned(16))) float ls[2 * NS * XP_LSP];
    const int tid = threadIdx.x, lane = tid & 63, wave = wave_id(tid);
#if defined(XP_TRACE)
    __shared__ unsigned long long tr[128 * 4];
    const int tr_slot = blockIdx.x == 7 ? 0 : (blockIdx.x == gridDim.x / 2 + 3 ? 1 : -1);
    const bool tr_on = tr_slot >= 0 && threadIdx.x == 0 && dir == 0;
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memtime();
#endif
    if (prio == 1) {  // static issue priority for one of the two workgroups that share a CU (2 / 3: per phase, in the step)
        const unsigned hw_id = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID[3:0] = wave slot
        if (hw_id & 1) __builtin_amdgcn_s_setprio(2);
    }
#if defined(LH_PROBE_INTRA1)       // timing probe build only (WRONG results: the reverse tiles race the forward ones): both directions of
    const int nt1 = (nseq + NS - 1) / NS;      // lh_intra_block in ONE launch, forward tiles first (what would it buy?)
    dir = (int)blockIdx.x >= nt1;
    accumulate = dir;
    wlin_pk += (long)dir * 4 * 2 * 64 * 16;
    const int s0 = ((int)blockIdx.x - dir * nt1) * NS;
#else
    const int s0 = blockIdx.x * NS;
#endif
    const int g4 = lane >> 4, l15 = lane & 15;
    const int q = tid & 15;
    const int unit = wave * 16 + l15;

    // addressing as in k_ln_lstm_lin: uniform 64-bit step base (scalar unit) + one 32-bit per-thread byte offset
    auto row_of0 = [&](int s) -> long { return (long)(s / sdiv) * so + (long)(s % sdiv) * si; };
    const long wg_row0 = row_of0(min(s0, nseq - 1));
    const int srow = min(s0 + (tid >> 4), nseq - 1);                      // tail rows replicate sequence nseq-1
    const unsigned voff = (unsigned)((row_of0(srow) - wg_row0) * (C * 4) + q * 16);
    const char* xb = reinterpret_cast<const char*>(x) + wg_row0 * (C * 4);
    char* ob = reinterpret_cast<char*>(out) + wg_row0 * (C * 4);
    const char* bsrc = reinterpret_cast<const char*>(
        accumulate ? reinterpret_cast<unsigned long long>(ob) : reinterpret_cast<unsigned long long>(xb));
    const long step_bytes = (long)ps * (C * 4);
    auto step_pos = [&](int it) -> int { it = min(max(it, 0), nstep - 1); return dir ? (nstep - 1 - it) : it; };

    // resident weights: gate image [dir][wave][gate][ks][lane][hi 8 | lo 8] (k-steps 0,1 = x half, 2,3 = h half) and this
    // wave's 16 output-projection columns
    xp_f16x8 wh[4][4], wl[4][4];
    {
        const _Float16* wp = w_pk + ((long)(dir * 4 + wave) * 16 * 64 + lane) * 16;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wh[g][ks] = *reinterpret_cast<const xp_f16x8*>(wp + (long)(g * 4 + ks) * 64 * 16);
                wl[g][ks] = *reinterpret_cast<const xp_f16x8*>(wp + (long)(g * 4 + ks) * 64 * 16 + 8);
            }
    }
    xp_f16x8 lwh[2], lwl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        lwh[ks] = *reinterpret_cast<const xp_f16x8*>(&wlin_pk[((wave * 2 + ks) * 64 + lane) * 16]);
        lwl[ks] = *reinterpret_cast<const xp_f16x8*>(&wlin_pk[((wave * 2 + ks) * 64 + lane) * 16 + 8]);
    }
    // gate bias as a factor: the packed bias carries the gate's exponent scale, 1 + 2^(a + b) = fma(2^a, 2^b, 1)
    float eb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)      // exponent clamped to +-100: 2^a may saturate to inf / 0 on its own, and inf * 0 would be NaN
        eb[g] = __builtin_amdgcn_exp2f(fminf(fmaxf(b_sum[dir * 256 + g * 64 + unit], -100.0f), 100.0f));
    const float lbias = accumulate ? 0.0f : blin[unit];

    const int a_row = (tid >> 4) * XP_AP + q * 4;      // row-wise role: row tid >> 4, float4 q
    const int l_row = (tid >> 4) * XP_LSP + q * 4;
    const int a_frag = l15 * XP_AP + g4 * 8;           // MFMA A fragment: row l15, halves g4*8 (+32 ks)
    const int a_cell = (g4 * 4) * XP_AP + C + unit;    // cell role: rows g4*4 + r, hidden column `unit`
    const int l_cell = (g4 * 4) * XP_LSP + unit;

    auto load_x = [&](int it) -> float4 { return *reinterpret_cast<const float4*>(xb + step_pos(it) * step_bytes + voff); };
    auto load_base = [&](int it) -> float4 { return *reinterpret_cast<const float4*>(bsrc + step_pos(it) * step_bytes + voff); };
    auto store_split4 = [&](int idx, float a, float b, float c, float d) {
        f16x2_t h01, l01, h23, l23;
        split_pair(a, b, h01, l01);
        split_pair(c, d, h23, l23);
        const xp_f16x4 h4 = xp_f16x4{h01[0], h01[1], h23[0], h23[1]}, l4 = xp_f16x4{l01[0], l01[1], l23[0], l23[1]};
        *reinterpret_cast<xp_f16x4*>(&ahi[idx]) = h4;
        *reinterpret_cast<xp_f16x4*>(&alo[idx]) = l4;
    };
    auto norm_store_x = [&](int buf, float4 v) {
        const float mean = group16_sum(v.x + v.y + v.z + v.w) * (1.0f / C);
        v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
        const float var = group16_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w) * (1.0f / C);
        const float rstd = __builtin_amdgcn_rsqf(var + LN_EPS);          // var + eps >= 1e-5: no denormal guard needed
        store_split4(buf * NS * XP_AP + a_row, v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd);
    };
    // non-recurrent half of one step's gate pre-activations from the x fragments (k-steps 0,1) of A buffer `buf`
    auto x_half = [&](int buf, f32x4 (&g)[4]) __attribute__((always_inline)) {
        xp_f16x8 xh[2], xl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            xh[ks] = *reinterpret_cast<const xp_f16x8*>(&ahi[buf * NS * XP_AP + a_frag + ks * 32]);
            xl[ks] = *reinterpret_cast<const xp_f16x8*>(&alo[buf * NS * XP_AP + a_frag + ks * 32]);
        }
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) g[gg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) g[gg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[ks], wh[gg][ks], g[gg], 0, 0, 0);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) g[gg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[ks], wl[gg][ks], g[gg], 0, 0, 0);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) g[gg] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[ks], wh[gg][ks], g[gg], 0, 0, 0);
        }
    };

    // ---- prologue: LN(x_0), LN(x_1) in the two buffers, x_2 in flight, h_{-1} = 0, x half of step 0
    float creg[4] = {0.f, 0.f, 0.f, 0.f};            // cell state, scaled by -2 log2 e
    float4 xr, rr = make_float4(0.f, 0.f, 0.f, 0.f);
    norm_store_x(0, load_x(0));
    norm_store_x(1, load_x(1));
    xr = load_x(2);
    store_split4(a_row + C, 0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    f32x4 gx[4];
    x_half(0, gx);
    __syncthreads();                                  // buffer 0's x half is rewritten (x_2) in step 0

    // STORE = false: the first two steps, which have no finished rows yet (peeled: a uniform branch around the store would
    // cut the step's scheduling region in two).
    // The step is written as two "zippers": slot i = one MFMA + the i-th slice of a list of small vector operations,
    // closed by a scheduling fence, so the issue order is the one written here (hipcc's own schedule of the same code
    // clusters the MFMAs in runs of 8..24 and leaves the vector work in slabs between them).
    auto step = [&](int it, auto cur_tag, auto store_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        constexpr bool STORE = decltype(store_tag)::value;
        // ================= phase H: recurrent half on top of gx  ||  row-wise work
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        if (prio == 3) __builtin_amdgcn_s_setprio(0);
        XP_STAMP(0);
        xp_f16x8 hh[2], hl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            hh[ks] = *reinterpret_cast<const xp_f16x8*>(&ahi[cur * NS * XP_AP + a_frag + (2 + ks) * 32]);
            hl[ks] = *reinterpret_cast<const xp_f16x8*>(&alo[cur * NS * XP_AP + a_frag + (2 + ks) * 32]);
        }
        f32x4 acc[4] = {gx[0], gx[1], gx[2], gx[3]};
        auto h_mfma = [&](auto idx) __attribute__((always_inline)) {
            constexpr int i = decltype(idx)::value, ks = i / 12, p = (i % 12) / 4, g = i % 4;
            if constexpr (p < XP_NP)
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? hl[ks] : hh[ks], p == 1 ? wl[g][2 + ks] : wh[g][2 + ks],
                                                            acc[g], 0, 0, 0);
        };
        // row-wise: x_{it+2} (fetched a step ago) normalised into the x half of buffer `cur` (its x_it was consumed by the
        // x half of the previous step), rows of step it-2 finished and stored, next loads
        float4 v, pv, done, y;
        float s, t, qa, qb, rstd;
        xp_f16x4 h4, l4;
        auto h_ops = std::make_tuple(
            [&] { v = xr; s = v.x + v.y; t = v.z + v.w; },
            [&] { s += t; pv = *reinterpret_cast<const float4*>(&ls[nxt * NS * XP_LSP + l_row]); },
            [&] { s = row_ror_add<8>(s); },
            [&] { s = row_ror_add<4>(s); },
            [&] { s = row_ror_add<2>(s); },
            [&] { s = row_ror_add<1>(s); },
            [&] { v.x = __builtin_fmaf(s, -1.0f / C, v.x); v.y = __builtin_fmaf(s, -1.0f / C, v.y); },
            [&] { v.z = __builtin_fmaf(s, -1.0f / C, v.z); v.w = __builtin_fmaf(s, -1.0f / C, v.w); },
            [&] { qa = v.x * v.x; qb = v.z * v.z; },
            [&] { qa = __builtin_fmaf(v.y, v.y, qa); qb = __builtin_fmaf(v.w, v.w, qb); },
            [&] { qa += qb; done.x = rr.x + pv.x; },
            [&] { qa = row_ror_add<8>(qa); done.y = rr.y + pv.y; },
            [&] { qa = row_ror_add<4>(qa); done.z = rr.z + pv.z; },
            [&] { qa = row_ror_add<2>(qa); done.w = rr.w + pv.w; },
            [&] { qa = row_ror_add<1>(qa); },
            [&] {
                rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(qa, 1.0f / C, LN_EPS));   // var + eps >= 1e-5: no denormal guard
#if defined(XP_PROBE_NOGLOBAL)
                if (STORE && done.x == 1.2345e30f) *reinterpret_cast<float4*>(ob + step_pos(it - 2) * step_bytes + voff) = done;
#else
                if (STORE) *reinterpret_cast<float4*>(ob + step_pos(it - 2) * step_bytes + voff) = done;
#endif
            },
#if defined(XP_PROBE_NOGLOBAL)
            [&] { if (!STORE) rr = load_base(it - 1); },
#else
            [&] { rr = load_base(it - 1); },
#endif
            [&] { y.x = v.x * rstd; y.y = v.y * rstd; },
            [&] { y.z = v.z * rstd; y.w = v.w * rstd; },
            [&] { f16x2_t a_, b_; split_pair(y.x, y.y, a_, b_); h4[0] = a_[0]; h4[1] = a_[1]; l4[0] = b_[0]; l4[1] = b_[1]; },
            [&] { },
            [&] { f16x2_t a_, b_; split_pair(y.z, y.w, a_, b_); h4[2] = a_[0]; h4[3] = a_[1]; l4[2] = b_[0]; l4[3] = b_[1]; },
            [&] {
                *reinterpret_cast<xp_f16x4*>(&ahi[cur * NS * XP_AP + a_row]) = h4;
                *reinterpret_cast<xp_f16x4*>(&alo[cur * NS * XP_AP + a_row]) = l4;
            },
#if defined(XP_PROBE_NOGLOBAL)
            [&] { if (!STORE) xr = load_x(it + 3); });
#else
            [&] { xr = load_x(it + 3); });
#endif
        xp_zip<24>(h_mfma, h_ops);
        XP_STAMP(1);
        if (prio == 2) __builtin_amdgcn_s_setprio(0);
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        // ================= phase C: projection of h_{it-1} (6 MFMAs), x half of step it+1 (24)  ||  cell update of step it
        xp_f16x8 xh[2], xl[2];
        f32x4 am = f32x4{lbias, lbias, lbias, lbias};
        auto c_mfma = [&](auto idx) __attribute__((always_inline)) {
            constexpr int i = decltype(idx)::value;
            if constexpr (i < 6) {
                constexpr int ks = i / 3, p = i % 3;
                if constexpr (p < XP_NP)
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? hl[ks] : hh[ks], p == 1 ? lwl[ks] : lwh[ks], am, 0, 0, 0);
            } else {
                constexpr int j = i - 6, ks = j / XP_XN, p = (j % XP_XN) / 4, g = j % 4;
                const f32x4 c0 = j < 4 ? f32x4{0.f, 0.f, 0.f, 0.f} : gx[g];
                if constexpr (p < XP_NP)
                gx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? xl[ks] : xh[ks], p == 1 ? wl[g][ks] : wh[g][ks], c0, 0, 0, 0);
            }
        };
        float tc[4];
        // cell r of this lane, in place in the accumulators: acc[g][r] -> 2^a -> 1 + 2^(a+b) -> gate value
        auto cA = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[0][r] = xp_exp2(acc[0][r]); acc[1][r] = xp_exp2(acc[1][r]); };
        auto cB = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[2][r] = xp_exp2(acc[2][r]); acc[3][r] = xp_exp2(acc[3][r]); };
        auto cC = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g][r] = __builtin_fmaf(acc[g][r], eb[g], 1.0f);
        };
        auto cD = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[0][r] = xp_rcp(acc[0][r]); acc[1][r] = xp_rcp(acc[1][r]); };
        auto cE = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[2][r] = xp_rcp(acc[2][r]); acc[3][r] = xp_rcp(acc[3][r]); };
        auto cF = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const float g2 = __builtin_fmaf(2.0f * XP_K2, acc[2][r], -XP_K2);          // -2 log2e * tanh(g)
            creg[r] = __builtin_fmaf(acc[1][r], creg[r], acc[0][r] * g2);              // scaled cell state
        };
        auto cG = [&](auto r_) { constexpr int r = decltype(r_)::value; tc[r] = 1.0f + xp_exp2(creg[r]); };
        auto cH = [&](auto r_) { constexpr int r = decltype(r_)::value; tc[r] = xp_rcp(tc[r]); };
        auto cI = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const float hv = acc[3][r] * __builtin_fmaf(2.0f, tc[r], -1.0f);
            _Float16 th, tl;
            split_hl(hv, th, tl);
            ahi[(nxt * NS + r) * XP_AP + a_cell] = th;
            alo[(nxt * NS + r) * XP_AP + a_cell] = tl;
        };
        using R0 = std::integral_constant<int, 0>;
        using R1 = std::integral_constant<int, 1>;
        using R2 = std::integral_constant<int, 2>;
        using R3 = std::integral_constant<int, 3>;
        auto c_ops = std::make_tuple(
            [&] {
                xh[0] = *reinterpret_cast<const xp_f16x8*>(&ahi[nxt * NS * XP_AP + a_frag]);
                xl[0] = *reinterpret_cast<const xp_f16x8*>(&alo[nxt * NS * XP_AP + a_frag]);
            },
            [&] {
                xh[1] = *reinterpret_cast<const xp_f16x8*>(&ahi[nxt * NS * XP_AP + a_frag + 32]);
                xl[1] = *reinterpret_cast<const xp_f16x8*>(&alo[nxt * NS * XP_AP + a_frag + 32]);
            },
            [&] { cA(R0{}); }, [&] { cA(R1{}); }, [&] { cB(R0{}); }, [&] { cB(R1{}); }, [&] { cC(R0{}); }, [&] { cC(R1{}); },
            [&] { cD(R0{}); }, [&] { cD(R1{}); },
            [&] {
#pragma unroll
                for (int r = 0; r < 4; ++r) ls[(cur * NS + r) * XP_LSP + l_cell] = am[r];
            },
            [&] { cE(R0{}); }, [&] { cE(R1{}); }, [&] { cF(R0{}); }, [&] { cF(R1{}); }, [&] { cG(R0{}); }, [&] { cG(R1{}); },
            [&] { cH(R0{}); }, [&] { cH(R1{}); }, [&] { cI(R0{}); }, [&] { cI(R1{}); },
            [&] { cA(R2{}); }, [&] { cA(R3{}); }, [&] { cB(R2{}); }, [&] { cB(R3{}); }, [&] { cC(R2{}); }, [&] { cC(R3{}); },
            [&] { cD(R2{}); }, [&] { cD(R3{}); }, [&] { cE(R2{}); }, [&] { cE(R3{}); }, [&] { cF(R2{}); }, [&] { cF(R3{}); },
            [&] { cG(R2{}); }, [&] { cG(R3{}); }, [&] { cH(R2{}); }, [&] { cH(R3{}); }, [&] { cI(R2{}); }, [&] { cI(R3{}); });
        xp_zip<6 + 2 * XP_XN>(c_mfma, c_ops);
        XP_STAMP(2);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
    };
    {
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        step(0, B0{}, std::false_type{});
        if (nstep > 1) step(1, B1{}, std::false_type{});
        int it = 2;
        for (; it + 1 < nstep; it += 2) {
            step(it, B0{}, std::true_type{});
            step(it + 1, B1{}, std::true_type{});
        }
        if (it < nstep) step(it, B0{}, std::true_type{});
    }

#if defined(XP_TRACE)
    if (tr_slot >= 0 && dir == 0 && threadIdx.x < 64)
        for (int i = threadIdx.x; i < 128 * 4; i += 64) xp_trace_buf[tr_slot * 512 + i] = tr[i];
    if (threadIdx.x == 0 && dir == 0 && blockIdx.x < 4096) {
        xp_wg_log[blockIdx.x * 4 + 0] = wg_t0;
        xp_wg_log[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memtime();
        xp_wg_log[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
        xp_wg_log[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 20);
    }
#endif
    // ---- drain: rows of the last two steps (projection of h_{nstep-1} still to do)
    const int lastb = nstep & 1;
    if (nstep >= 2) {
        const float4 pv = *reinterpret_cast<const float4*>(&ls[(lastb ^ 1) * NS * XP_LSP + l_row]);
        *reinterpret_cast<float4*>(ob + step_pos(nstep - 2) * step_bytes + voff) =
            make_float4(rr.x + pv.x, rr.y + pv.y, rr.z + pv.z, rr.w + pv.w);
    }
    rr = load_base(nstep - 1);
    {
        f32x4 am = f32x4{lbias, lbias, lbias, lbias};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const xp_f16x8 ah = *reinterpret_cast<const xp_f16x8*>(&ahi[lastb * NS * XP_AP + a_frag + (2 + ks) * 32]);
            const xp_f16x8 al = *reinterpret_cast<const xp_f16x8*>(&alo[lastb * NS * XP_AP + a_frag + (2 + ks) * 32]);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwh[ks], am, 0, 0, 0);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwl[ks], am, 0, 0, 0);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, lwh[ks], am, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ls[(lastb * NS + r) * XP_LSP + l_cell] = am[r];
    }
    __syncthreads();
    {
        const float4 pv = *reinterpret_cast<const float4*>(&ls[lastb * NS * XP_LSP + l_row]);
        *reinterpret_cast<float4*>(ob + step_pos(nstep - 1) * step_bytes + voff) =
            make_float4(rr.x + pv.x, rr.y + pv.y, rr.z + pv.z, rr.w + pv.w);
    }
}

// ------------------------------------------------------------------------------------------------------
// k_inter_xp: the inter pass (LayerNorm + causal LSTM over time with carried (h0, c0) + Linear + residual,
// tfgridnet_causal.py:521-538) with the same hand-ordered step.  Tile, LDS images, weight images and roles are those of
// k_lstm_lin8p (lh_lstm.hip): 16 sequences x all steps per 8-wave workgroup, one per CU; transposed gate GEMM (weights =
// MFMA A operand, rows ordered (unit, gate)) so a lane holds the four gates of two cells; waves 0..3 ("LIN") also run the
// output projection and finish / fetch the output rows, waves 4..7 normalise and split x; the x half of step t+1 is
// computed during step t.  Per step and wave:
//   phase H:  12 MFMAs (h half, on the chain)                          ||  the wave's row-wise role
//   phase C:  12 MFMAs (x half of step t+1) [+ 6 projection, LIN]      ||  the two cell updates (20 transcendentals)
// ------------------------------------------------------------------------------------------------------
template <bool LIN>
struct xp_tag { static constexpr bool value = LIN; };

__global__ void __launch_bounds__(512, 1) k_inter_xp(const float* __restrict__ x, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ b_sum, const _Float16* __restrict__ wlin_pk,
                                                     const float* __restrict__ blin, const float* __restrict__ h0,
                                                     const float* __restrict__ c0, float* __restrict__ hN,
                                                     float* __restrict__ cN, float* out, int nseq, int nstep, int sdiv,
                                                     int so, int si, int ps, int dir, int accumulate, int prio, int cflags) {
    // cflags (time windows, lh_inter_block_win `carry`): bit 0 = c0 already holds the kernel's scaled cell state -2 log2(e) c
    // (written by the previous window with bit 1), bit 1 = cN is written that way — the state then crosses a window boundary
    // without the two roundings of c / k and k * c, and the windows reproduce the whole-clip launch bit for bit
    constexpr int NS = 16;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * NS * XP_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * NS * XP_AP];
    __shared__ __attribute__((aligned(16))) float ls[2 * NS * XP_LSP];
    __shared__ __attribute__((aligned(16))) float hf[NS * XP_LSP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Roles: four waves run the output projection and finish rows ("lin"), four normalise and split x ("LN").  The two waves
    // of a SIMD are w and w + 4, and without a hint the lower-numbered one wins the arbiter; `prio` bit 4 (lh_set_tuning key 9,
    // + 16) swaps the roles' halves: LN — the heavier on-chain phase in the s_memtime trace — on waves 0..3.
    const bool role_swap = (prio & 16) != 0;
    prio &= 15;
    const bool lin_wave = (wave < 4) != role_swap;
    const int rw = wave & 3;                      // index inside the role's four waves
    // A/B switch (lh_set_tuning key 9): issue priority for one role — 1 = the LayerNorm waves, 2 = the projection waves
    if ((prio == 1 && !lin_wave) || (prio == 2 && lin_wave)) __builtin_amdgcn_s_setprio(3);
#if defined(XP_TRACE)
    __shared__ unsigned long long tr_all[2 * 128 * 4];
    const bool tr_on = blockIdx.x == 3 && (threadIdx.x & 255) == 0;
    unsigned long long* const tr = tr_all + (threadIdx.x >> 8) * 512;
#endif
    const int s0 = blockIdx.x * NS;
    const int g4 = lane >> 4, l15 = lane & 15;
    const int q = tid & 15, rrow = (tid & 255) >> 4;
    // a lane's two cells are ADJACENT units 8w + 2 g4 + m (tile m holds the even / odd units of the wave: weights.py
    // pack_lstm_f16x3_w8), so h leaves as ONE 4-byte LDS store per half instead of two 2-byte stores 8 bytes apart (the 2-byte
    // stores were 4-way bank conflicts on the 288-byte row stride: SQ_LDS_BANK_CONFLICT 0.34 of the LDS cycles in round 3)
    const int unit0 = 8 * wave + 2 * g4;

    auto row_of0 = [&](int s) -> long { return (long)(s / sdiv) * so + (long)(s % sdiv) * si; };
    const long wg_row0 = row_of0(min(s0, nseq - 1));
    const unsigned voff = (unsigned)((row_of0(min(s0 + rrow, nseq - 1)) - wg_row0) * (C * 4) + q * 16);
    const char* xb = reinterpret_cast<const char*>(x) + wg_row0 * (C * 4);
    char* ob = reinterpret_cast<char*>(out) + wg_row0 * (C * 4);
    const char* bsrc = reinterpret_cast<const char*>(
        accumulate ? reinterpret_cast<unsigned long long>(ob) : reinterpret_cast<unsigned long long>(xb));
    const long step_bytes = (long)ps * (C * 4);
    auto step_pos = [&](int it) -> int { it = min(max(it, 0), nstep - 1); return dir ? (nstep - 1 - it) : it; };

    xp_f16x8 wh[2][4], wl[2][4];          // image [dir][wave][tile m][ks][lane][hi 8 | lo 8] (weights.py pack_lstm_f16x3_w8)
    {
        const _Float16* wp = w_pk + ((long)(dir * 8 + wave) * 8 * 64 + lane) * 16;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                wh[m][ks] = *reinterpret_cast<const xp_f16x8*>(wp + (long)(m * 4 + ks) * 64 * 16);
                wl[m][ks] = *reinterpret_cast<const xp_f16x8*>(wp + (long)(m * 4 + ks) * 64 * 16 + 8);
            }
    }
    xp_f16x8 lwh[2], lwl[2];
    float lbias = 0.0f;
    if (lin_wave) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            lwh[ks] = *reinterpret_cast<const xp_f16x8*>(&wlin_pk[((rw * 2 + ks) * 64 + lane) * 16]);
            lwl[ks] = *reinterpret_cast<const xp_f16x8*>(&wlin_pk[((rw * 2 + ks) * 64 + lane) * 16 + 8]);
        }
        if (!accumulate) lbias = blin[rw * 16 + l15];
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lwh[ks] = lwl[ks] = xp_f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    float eb[2][4];                       // 2^bias: the gate bias enters the cell update as a factor (see k_intra_xp)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g)      // exponent clamped like k_intra_xp's (no inf * 0)
            eb[m][g] = __builtin_amdgcn_exp2f(fminf(fmaxf(b_sum[dir * 256 + g * 64 + unit0 + m], -100.0f), 100.0f));

    const int a_frag = l15 * XP_AP + g4 * 8;
    const int a_cell = l15 * XP_AP + C + unit0;
    const int a_row = rrow * XP_AP + q * 4;
    const int l_row = rrow * XP_LSP + q * 4;
    const int l_lin = (g4 * 4) * XP_LSP + rw * 16 + l15;

    auto store_split4 = [&](int idx, float a, float b, float c, float d) {
        f16x2_t h01, l01, h23, l23;
        split_pair(a, b, h01, l01);
        split_pair(c, d, h23, l23);
        const xp_f16x4 h4 = xp_f16x4{h01[0], h01[1], h23[0], h23[1]}, l4 = xp_f16x4{l01[0], l01[1], l23[0], l23[1]};
        *reinterpret_cast<xp_f16x4*>(&ahi[idx]) = h4;
        *reinterpret_cast<xp_f16x4*>(&alo[idx]) = l4;
    };
    // the first two rows of a launch, in EXACTLY the arithmetic of the step's row-wise role below (same sums, same fused
    // operations): a time window (lh_inter_block_win) starts here with rows that the whole-clip launch normalises inside its
    // loop, and the two must agree bit for bit
    auto norm_store_x = [&](int buf, float4 v) {
        float s = v.x + v.y, t = v.z + v.w;
        s += t;
        s = group16_sum(s);
        v.x = __builtin_fmaf(s, -1.0f / C, v.x); v.y = __builtin_fmaf(s, -1.0f / C, v.y);
        v.z = __builtin_fmaf(s, -1.0f / C, v.z); v.w = __builtin_fmaf(s, -1.0f / C, v.w);
        float qa = v.x * v.x, qb = v.z * v.z;
        qa = __builtin_fmaf(v.y, v.y, qa); qb = __builtin_fmaf(v.w, v.w, qb);
        qa += qb;
        qa = group16_sum(qa);
        const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(qa, 1.0f / C, LN_EPS));
        store_split4(buf * NS * XP_AP + a_row, v.x * rstd, v.y * rstd, v.z * rstd, v.w * rstd);
    };
    auto load_row = [&](const char* base, int it) -> float4 {
        return *reinterpret_cast<const float4*>(base + step_pos(it) * step_bytes + voff);
    };
    auto x_half = [&](int buf, f32x4 (&g)[2]) {
#pragma unroll
        for (int m = 0; m < 2; ++m) g[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const xp_f16x8 xh = *reinterpret_cast<const xp_f16x8*>(&ahi[buf * NS * XP_AP + a_frag + ks * 32]);
            const xp_f16x8 xl = *reinterpret_cast<const xp_f16x8*>(&alo[buf * NS * XP_AP + a_frag + ks * 32]);
#pragma unroll
            for (int m = 0; m < 2; ++m) g[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[m][ks], xh, g[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) g[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[m][ks], xl, g[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m) g[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[m][ks], xh, g[m], 0, 0, 0);
        }
    };

    // ---- prologue: x_0, x_1 normalised into the two buffers, x_2 in flight; h_{-1}; x half of step 0
    float creg[2], hreg[2];               // creg: cell state scaled by -2 log2 e
    // global rows are fetched XP_PF steps ahead of their use (one register quad per step in flight)
    float4 carryq[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (!lin_wave) {
        norm_store_x(0, load_row(xb, 0));
        norm_store_x(1, load_row(xb, 1));
        carryq[0] = load_row(xb, 2);
        if (XP_PF == 2) carryq[1] = load_row(xb, 3);
    } else {
        const int s = min(s0 + rrow, nseq - 1);
        float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h0) hv = *reinterpret_cast<const float4*>(&h0[(long)s * H + q * 4]);
        store_split4(a_row + C, hv.x, hv.y, hv.z, hv.w);
    }
    {
        const int s = min(s0 + l15, nseq - 1);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            creg[m] = c0 ? ((cflags & 1) ? 1.0f : XP_K2) * c0[(long)s * H + unit0 + m] : 0.0f;
            hreg[m] = 0.0f;
        }
    }
    __syncthreads();
    f32x4 gx[2];
    x_half(0, gx);
    __syncthreads();                                  // buffer 0's x half is rewritten (x_2) in step 0

    auto step = [&](int it, auto cur_tag, auto lin_tag, auto store_tag) __attribute__((always_inline)) {
        constexpr int cur = decltype(cur_tag)::value, nxt = cur ^ 1;
        constexpr bool LIN = decltype(lin_tag)::value;
        constexpr bool STORE = decltype(store_tag)::value;
        float4& carry = carryq[XP_PF == 2 ? cur : 0];
        if (prio == 3) __builtin_amdgcn_s_setprio(3);
        if (prio == 4) __builtin_amdgcn_s_setprio(0);
        XP_STAMP(0);
        // ================= phase H: acc = (x half of this step, computed a step ago) + h_{it-1} W_hh^T  ||  row-wise role
        xp_f16x8 bh[2], bl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bh[ks] = *reinterpret_cast<const xp_f16x8*>(&ahi[cur * NS * XP_AP + a_frag + (2 + ks) * 32]);
            bl[ks] = *reinterpret_cast<const xp_f16x8*>(&alo[cur * NS * XP_AP + a_frag + (2 + ks) * 32]);
        }
        f32x4 acc[2] = {gx[0], gx[1]};
        xp_f16x8 xh[2], xl[2];            // x fragments of step it+1 (buffer nxt): fetched ahead of phase C's first MFMA
        auto h_mfma = [&](auto idx) __attribute__((always_inline)) {
            constexpr int i = decltype(idx)::value, ks = i / 6, p = (i % 6) / 2, m = i % 2;
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? wl[m][2 + ks] : wh[m][2 + ks], p == 1 ? bl[ks] : bh[ks],
                                                            acc[m], 0, 0, 0);
        };
        if constexpr (LIN) {
            // rows of step it-2: base (fetched a step ago) + projection parked in ls[nxt]; then this step's base fetch
            float4 pv, done;
            auto h_ops = std::make_tuple(
                [&] { pv = *reinterpret_cast<const float4*>(&ls[nxt * NS * XP_LSP + l_row]); },
                [&] {}, [&] {}, [&] {},
                [&] { done.x = carry.x + pv.x; done.y = carry.y + pv.y; },
                [&] { done.z = carry.z + pv.z; done.w = carry.w + pv.w; },
                [&] { if (STORE) *reinterpret_cast<float4*>(ob + step_pos(it - 2) * step_bytes + voff) = done; },
                [&] { carry = load_row(bsrc, it + XP_PF - 2); });
            xp_zip<12>(h_mfma, h_ops);
        } else {
            // x_{it+2} (fetched a step ago) normalised into the x half of buffer `cur`; x_{it+3} fetched
            float4 v, y;
            float s, t, qa, qb, rstd;
            xp_f16x4 h4, l4;
            auto h_ops = std::make_tuple(
                [&] { v = carry; s = v.x + v.y; t = v.z + v.w; },
                [&] { s += t; },
                [&] { s = row_ror_add<8>(s); },
                [&] { s = row_ror_add<4>(s); },
                [&] { s = row_ror_add<2>(s); },
                [&] { s = row_ror_add<1>(s); },
                [&] { v.x = __builtin_fmaf(s, -1.0f / C, v.x); v.y = __builtin_fmaf(s, -1.0f / C, v.y); },
                [&] { v.z = __builtin_fmaf(s, -1.0f / C, v.z); v.w = __builtin_fmaf(s, -1.0f / C, v.w); },
                [&] { qa = v.x * v.x; qb = v.z * v.z; },
                [&] { qa = __builtin_fmaf(v.y, v.y, qa); qb = __builtin_fmaf(v.w, v.w, qb); },
                [&] { qa += qb; },
                [&] { qa = row_ror_add<8>(qa); },
                [&] { qa = row_ror_add<4>(qa); },
                [&] { qa = row_ror_add<2>(qa); },
                [&] { qa = row_ror_add<1>(qa); },
                [&] { rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(qa, 1.0f / C, LN_EPS)); },
                [&] { y.x = v.x * rstd; y.y = v.y * rstd; },
                [&] { y.z = v.z * rstd; y.w = v.w * rstd; },
                [&] { carry = load_row(xb, it + 2 + XP_PF); },      // behind the last use of v: the fetch can land in v's registers
                [&] { f16x2_t a_, b_; split_pair(y.x, y.y, a_, b_); h4[0] = a_[0]; h4[1] = a_[1]; l4[0] = b_[0]; l4[1] = b_[1]; },
                [&] { },
                [&] { f16x2_t a_, b_; split_pair(y.z, y.w, a_, b_); h4[2] = a_[0]; h4[3] = a_[1]; l4[2] = b_[0]; l4[3] = b_[1]; },
                [&] {
                    *reinterpret_cast<xp_f16x4*>(&ahi[cur * NS * XP_AP + a_row]) = h4;
                    *reinterpret_cast<xp_f16x4*>(&alo[cur * NS * XP_AP + a_row]) = l4;
                },
                [&] {
                    xh[0] = *reinterpret_cast<const xp_f16x8*>(&ahi[nxt * NS * XP_AP + a_frag]);
                    xl[0] = *reinterpret_cast<const xp_f16x8*>(&alo[nxt * NS * XP_AP + a_frag]);
                    xh[1] = *reinterpret_cast<const xp_f16x8*>(&ahi[nxt * NS * XP_AP + a_frag + 32]);
                    xl[1] = *reinterpret_cast<const xp_f16x8*>(&alo[nxt * NS * XP_AP + a_frag + 32]);
                });
            xp_zip<12>(h_mfma, h_ops);
        }
        XP_STAMP(1);
        if (prio == 3) __builtin_amdgcn_s_setprio(0);
        if (prio == 4) __builtin_amdgcn_s_setprio(3);
        // ================= phase C: [projection of h_{it-1},] x half of step it+1  ||  the two cell updates of step it
        f32x4 am = f32x4{lbias, lbias, lbias, lbias};
        constexpr int NL = LIN ? 6 : 0;
        auto c_mfma = [&](auto idx) __attribute__((always_inline)) {
            constexpr int i = decltype(idx)::value;
            if constexpr (i < NL) {
                constexpr int ks = i / 3, p = i % 3;
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? bl[ks] : bh[ks], p == 1 ? lwl[ks] : lwh[ks], am, 0, 0, 0);
            } else {
                constexpr int j = i - NL, ks = j / 6, p = (j % 6) / 2, m = j % 2;
                const f32x4 c0v = j < 2 ? f32x4{0.f, 0.f, 0.f, 0.f} : gx[m];
                gx[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? wl[m][ks] : wh[m][ks], p == 1 ? xl[ks] : xh[ks], c0v, 0, 0, 0);
            }
        };
        float tc[2];
        auto cA = [&](auto m_) { constexpr int m = decltype(m_)::value; acc[m][0] = __builtin_amdgcn_exp2f(acc[m][0]); acc[m][1] = __builtin_amdgcn_exp2f(acc[m][1]); };
        auto cB = [&](auto m_) { constexpr int m = decltype(m_)::value; acc[m][2] = __builtin_amdgcn_exp2f(acc[m][2]); acc[m][3] = __builtin_amdgcn_exp2f(acc[m][3]); };
        auto cC = [&](auto m_) {
            constexpr int m = decltype(m_)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[m][g] = __builtin_fmaf(acc[m][g], eb[m][g], 1.0f);
        };
        auto cD = [&](auto m_) { constexpr int m = decltype(m_)::value; acc[m][0] = __builtin_amdgcn_rcpf(acc[m][0]); acc[m][1] = __builtin_amdgcn_rcpf(acc[m][1]); };
        auto cE = [&](auto m_) { constexpr int m = decltype(m_)::value; acc[m][2] = __builtin_amdgcn_rcpf(acc[m][2]); acc[m][3] = __builtin_amdgcn_rcpf(acc[m][3]); };
        auto cF = [&](auto m_) {
            constexpr int m = decltype(m_)::value;
            const float g2 = __builtin_fmaf(2.0f * XP_K2, acc[m][2], -XP_K2);
            creg[m] = __builtin_fmaf(acc[m][1], creg[m], acc[m][0] * g2);
        };
        auto cG = [&](auto m_) { constexpr int m = decltype(m_)::value; tc[m] = 1.0f + __builtin_amdgcn_exp2f(creg[m]); };
        auto cH = [&](auto m_) { constexpr int m = decltype(m_)::value; tc[m] = __builtin_amdgcn_rcpf(tc[m]); };
        auto cI = [&](auto m_) {
            constexpr int m = decltype(m_)::value;
            hreg[m] = acc[m][3] * __builtin_fmaf(2.0f, tc[m], -1.0f);
            if constexpr (m == 1) {                // units unit0, unit0 + 1: one pair split, one packed store per half
                f16x2_t th2, tl2;
                split_pair(hreg[0], hreg[1], th2, tl2);
                *reinterpret_cast<f16x2_t*>(&ahi[nxt * NS * XP_AP + a_cell]) = th2;
                *reinterpret_cast<f16x2_t*>(&alo[nxt * NS * XP_AP + a_cell]) = tl2;
            }
        };
        using M0 = std::integral_constant<int, 0>;
        using M1 = std::integral_constant<int, 1>;
        auto c_ops = std::make_tuple(
            [&] {
                if constexpr (LIN) {          // (the LN waves fetched them at the end of phase H: their first MFMA here needs them)
                    xh[0] = *reinterpret_cast<const xp_f16x8*>(&ahi[nxt * NS * XP_AP + a_frag]);
                    xl[0] = *reinterpret_cast<const xp_f16x8*>(&alo[nxt * NS * XP_AP + a_frag]);
                }
            },
            [&] {
                if constexpr (LIN) {
                    xh[1] = *reinterpret_cast<const xp_f16x8*>(&ahi[nxt * NS * XP_AP + a_frag + 32]);
                    xl[1] = *reinterpret_cast<const xp_f16x8*>(&alo[nxt * NS * XP_AP + a_frag + 32]);
                }
            },
            [&] { cA(M0{}); }, [&] { cA(M1{}); }, [&] { cB(M0{}); }, [&] { cB(M1{}); }, [&] { cC(M0{}); }, [&] { cC(M1{}); },
            [&] { cD(M0{}); }, [&] { cD(M1{}); }, [&] { cE(M0{}); }, [&] { cE(M1{}); },
            [&] {
                if constexpr (LIN) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) ls[cur * NS * XP_LSP + r * XP_LSP + l_lin] = am[r];
                }
            },
            [&] { cF(M0{}); }, [&] { cF(M1{}); }, [&] { cG(M0{}); }, [&] { cG(M1{}); }, [&] { cH(M0{}); }, [&] { cH(M1{}); },
            [&] { cI(M0{}); }, [&] { cI(M1{}); });
        xp_zip<12 + NL>(c_mfma, c_ops);
        XP_STAMP(2);
        __syncthreads();
        XP_STAMP(3);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto run = [&](auto lin_tag) __attribute__((always_inline)) {
        using B0 = std::integral_constant<int, 0>;
        using B1 = std::integral_constant<int, 1>;
        step(0, B0{}, lin_tag, std::false_type{});
        step(1, B1{}, lin_tag, std::false_type{});        // nstep >= 2 (launch_inter_xp): one way into the loop, exact vmcnt
        int it = 2;
        for (; it + 1 < nstep; it += 2) {
            step(it, B0{}, lin_tag, std::true_type{});
            step(it + 1, B1{}, lin_tag, std::true_type{});
        }
        if (it < nstep) step(it, B0{}, lin_tag, std::true_type{});
    };
    if (lin_wave) run(std::true_type{});
    else run(std::false_type{});

#if defined(XP_TRACE)
    __syncthreads();
    if (blockIdx.x == 3)
        for (int i = threadIdx.x; i < 2 * 128 * 4; i += 512) xp_trace_inter[i] = tr_all[i];
#endif
    // ---- drain: rows of the last two steps (projection of h_{nstep-1} still to do), final state
    const int lastb = nstep & 1;
    float4 carry = XP_PF == 2 ? (lastb ? carryq[1] : carryq[0]) : carryq[0];      // base row of step nstep-2
    if (lin_wave) {
        if (nstep >= 2) {
            const float4 pv = *reinterpret_cast<const float4*>(&ls[(lastb ^ 1) * NS * XP_LSP + l_row]);
            *reinterpret_cast<float4*>(ob + step_pos(nstep - 2) * step_bytes + voff) =
                make_float4(carry.x + pv.x, carry.y + pv.y, carry.z + pv.z, carry.w + pv.w);
        }
        carry = XP_PF == 2 ? (lastb ? carryq[0] : carryq[1]) : load_row(bsrc, nstep - 1);
        f32x4 am = f32x4{lbias, lbias, lbias, lbias};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const xp_f16x8 ah = *reinterpret_cast<const xp_f16x8*>(&ahi[lastb * NS * XP_AP + a_frag + (2 + ks) * 32]);
            const xp_f16x8 al = *reinterpret_cast<const xp_f16x8*>(&alo[lastb * NS * XP_AP + a_frag + (2 + ks) * 32]);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwh[ks], am, 0, 0, 0);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, lwl[ks], am, 0, 0, 0);
            am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, lwh[ks], am, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) ls[lastb * NS * XP_LSP + r * XP_LSP + l_lin] = am[r];
    }
    if (hN) {
#pragma unroll
        for (int m = 0; m < 2; ++m) hf[l15 * XP_LSP + unit0 + m] = hreg[m];
    }
    __syncthreads();
    if (lin_wave) {
        const float4 pv = *reinterpret_cast<const float4*>(&ls[lastb * NS * XP_LSP + l_row]);
        *reinterpret_cast<float4*>(ob + step_pos(nstep - 1) * step_bytes + voff) =
            make_float4(carry.x + pv.x, carry.y + pv.y, carry.z + pv.z, carry.w + pv.w);
    } else if (hN && s0 + rrow < nseq) {
        *reinterpret_cast<float4*>(&hN[(long)(s0 + rrow) * H + q * 4]) = *reinterpret_cast<const float4*>(&hf[l_row]);
    }
    if (cN && s0 + l15 < nseq) {
#pragma unroll
        for (int m = 0; m < 2; ++m) cN[(long)(s0 + l15) * H + unit0 + m] = creg[m] * ((cflags & 2) ? 1.0f : 1.0f / XP_K2);
    }
}

static int g_xp_lds_pad = 0;      // lh_set_tuning(7, bytes): dynamic LDS added to k_intra_xp launches (timing probe: 1 workgroup per CU)
// Issue priorities (s_setprio; A/B in profiles/r03h_issue_priority.txt).  Two waves share every SIMD and the arbiter decides whose
// instruction goes first; without a hint the wave on the recurrence's critical path loses about every other arbitration.
static int g_xp_prio = 1;         // lh_set_tuning(8, v): k_intra_xp, 0 = none, 1 = the wave in the odd hardware slot of a SIMD runs at
                                  // priority 2 (default), 2 / 3 = every wave at priority 3 during phase H / phase C
static int g_xp_prio_inter = 3;   // lh_set_tuning(9, v): k_inter_xp, 0 = none, priority 3 for 1 = the LayerNorm waves, 2 = the projection
                                  // waves, 3 = every wave during phase H (default), 4 = every wave during phase C

int launch_inter_xp(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin,
                    const float* h0, const float* c0, float* hN, float* cN, float* out, int nseq, int nstep, int sdiv, int so,
                    int si, int ps, hipStream_t st, int cflags) {
    if (nstep < 2) return LH_ERR_ARG;             // the first two steps are peeled unconditionally
    hipLaunchKernelGGL(k_inter_xp, dim3((nseq + 15) / 16), dim3(512), 0, st, x, (const _Float16*)w_pk, b_sum,
                       (const _Float16*)wlin_pk, blin, h0, c0, hN, cN, out, nseq, nstep, sdiv, so, si, ps, 0, 0, g_xp_prio_inter,
                       cflags);
    return check_launch();
}
int xp_set(int key, int v) {
    if (key == 7) g_xp_lds_pad = v < 0 ? 0 : v;
    if (key == 8) g_xp_prio = v;
    if (key == 9) g_xp_prio_inter = v;
    return LH_OK;
}

int launch_intra_xp(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin, float* out,
                    int nseq, int nstep, int sdiv, int so, int si, int ps, int dir, int accumulate, hipStream_t st) {
#if defined(LH_PROBE_INTRA1)
    if (dir == 1) return LH_OK;               // the forward call launched both directions (wlin_pk: its own pointer, + 0)
    hipLaunchKernelGGL(k_intra_xp, dim3(2 * ((nseq + 15) / 16)), dim3(256), g_xp_lds_pad, st, x, (const _Float16*)w_pk, b_sum,
                       (const _Float16*)wlin_pk, blin, out, nseq, nstep, sdiv, so, si, ps, dir, accumulate, g_xp_prio);
    return check_launch();
#endif
    hipLaunchKernelGGL(k_intra_xp, dim3((nseq + 15) / 16), dim3(256), g_xp_lds_pad, st, x, (const _Float16*)w_pk, b_sum,
                       (const _Float16*)wlin_pk, blin, out, nseq, nstep, sdiv, so, si, ps, dir, accumulate, g_xp_prio);
    return check_launch();
}

}  // namespace lh

#if defined(XP_TRACE)
extern "C" int lh_probe_xp_trace_inter_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::xp_trace_inter), sizeof(lh::xp_trace_inter)) == hipSuccess ? 0 : 1;
}
extern "C" int lh_probe_xp_trace_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::xp_trace_buf), sizeof(lh::xp_trace_buf)) == hipSuccess ? 0 : 1;
}
extern "C" int lh_probe_xp_wglog_read(unsigned long long* host_dst) {
    return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(lh::xp_wg_log), sizeof(lh::xp_wg_log)) == hipSuccess ? 0 : 1;
}
#endif
