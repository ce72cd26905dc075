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
// and the instruction order inside each phase is given to the scheduler explicitly (sched_group_barrier: one MFMA, then
// the vector instructions that fit in its shadow) instead of "12 MFMAs, then a slab of vector work".  The measured issue
// model (profiles/r02a_ubench_issue_model.txt) is: a 16x16x32 MFMA holds the matrix pipe 19 cycles but the wave's issue
// only ~7.5, two plain vector instructions (or one transcendental) behind it are free.
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

// sched_group_barrier masks (LLVM AMDGPU): 0x2 VALU (not MFMA, not transcendental), 0x8 MFMA, 0x20 VMEM read,
// 0x40 VMEM write, 0x100 DS read, 0x200 DS write, 0x400 transcendental
#define XP_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// compile-time loop / zipper helpers
template <class F, int... I>
__device__ __forceinline__ void xp_sf(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void xp_for(F&& f) { xp_sf(f, std::make_integer_sequence<int, N>{}); }
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

#ifndef XP_H_VALU
#define XP_H_VALU 2            // plain vector instructions behind each phase-H MFMA
#endif
#ifndef XP_C_VALU
#define XP_C_VALU 2            // plain vector instructions behind each phase-C MFMA
#endif
#ifndef XP_C_TRANS
#define XP_C_TRANS 1           // transcendentals behind each phase-C MFMA
#endif

__global__ void __launch_bounds__(256, 2) k_intra_xp(const float* __restrict__ x, const _Float16* __restrict__ w_pk,
                                                     const float* __restrict__ b_sum, const _Float16* __restrict__ wlin_pk,
                                                     const float* __restrict__ blin, float* out, int nseq, int nstep,
                                                     int sdiv, int so, int si, int ps, int dir, int accumulate) {
    constexpr int NS = 16;
    __shared__ __attribute__((aligned(16))) _Float16 ahi[2 * NS * XP_AP];
    __shared__ __attribute__((aligned(16))) _Float16 alo[2 * NS * XP_AP];
    __shared__ __attribute__((aligned(16))) float ls[2 * NS * XP_LSP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s0 = blockIdx.x * NS;
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
    for (int g = 0; g < 4; ++g) eb[g] = __builtin_amdgcn_exp2f(b_sum[dir * 256 + g * 64 + unit]);
    const float lbias = accumulate ? 0.0f : blin[unit];

    const int a_row = (tid >> 4) * XP_AP + q * 4;      // row-wise role: row tid >> 4, float4 q
    const int l_row = (tid >> 4) * XP_LSP + q * 4;
    const int a_frag = l15 * XP_AP + g4 * 8;           // MFMA A fragment: row l15, halves g4*8 (+32 ks)
    const int a_cell = (g4 * 4) * XP_AP + C + unit;    // cell role: rows g4*4 + r, hidden column `unit`
    const int l_cell = (g4 * 4) * XP_LSP + unit;

    auto load_x = [&](int it) -> float4 { return *reinterpret_cast<const float4*>(xb + step_pos(it) * step_bytes + voff); };
    auto load_base = [&](int it) -> float4 { return *reinterpret_cast<const float4*>(bsrc + step_pos(it) * step_bytes + voff); };
    auto store_split4 = [&](int idx, float a, float b, float c, float d) {
        xp_f16x4 h4, l4;
        const float v[4] = {a, b, c, d};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const _Float16 th = (_Float16)v[i];
            h4[i] = th;
            l4[i] = (_Float16)(v[i] - (float)th);
        }
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
        xp_f16x8 hh[2], hl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            hh[ks] = *reinterpret_cast<const xp_f16x8*>(&ahi[cur * NS * XP_AP + a_frag + (2 + ks) * 32]);
            hl[ks] = *reinterpret_cast<const xp_f16x8*>(&alo[cur * NS * XP_AP + a_frag + (2 + ks) * 32]);
        }
        f32x4 acc[4] = {gx[0], gx[1], gx[2], gx[3]};
        auto h_mfma = [&](auto idx) __attribute__((always_inline)) {
            constexpr int i = decltype(idx)::value, ks = i / 12, p = (i % 12) / 4, g = i % 4;
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
                if (STORE) *reinterpret_cast<float4*>(ob + step_pos(it - 2) * step_bytes + voff) = done;
            },
            [&] { rr = load_base(it - 1); },
            [&] { y.x = v.x * rstd; y.y = v.y * rstd; },
            [&] { y.z = v.z * rstd; y.w = v.w * rstd; },
            [&] { h4[0] = (_Float16)y.x; h4[1] = (_Float16)y.y; h4[2] = (_Float16)y.z; h4[3] = (_Float16)y.w; },
            [&] { l4[0] = (_Float16)(y.x - (float)h4[0]); l4[1] = (_Float16)(y.y - (float)h4[1]); },
            [&] { l4[2] = (_Float16)(y.z - (float)h4[2]); l4[3] = (_Float16)(y.w - (float)h4[3]); },
            [&] {
                *reinterpret_cast<xp_f16x4*>(&ahi[cur * NS * XP_AP + a_row]) = h4;
                *reinterpret_cast<xp_f16x4*>(&alo[cur * NS * XP_AP + a_row]) = l4;
            },
            [&] { xr = load_x(it + 3); });
        xp_zip<24>(h_mfma, h_ops);
        // ================= phase C: projection of h_{it-1} (6 MFMAs), x half of step it+1 (24)  ||  cell update of step it
        xp_f16x8 xh[2], xl[2];
        f32x4 am = f32x4{lbias, lbias, lbias, lbias};
        auto c_mfma = [&](auto idx) __attribute__((always_inline)) {
            constexpr int i = decltype(idx)::value;
            if constexpr (i < 6) {
                constexpr int ks = i / 3, p = i % 3;
                am = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? hl[ks] : hh[ks], p == 1 ? lwl[ks] : lwh[ks], am, 0, 0, 0);
            } else {
                constexpr int j = i - 6, ks = j / 12, p = (j % 12) / 4, g = j % 4;
                const f32x4 c0 = j < 4 ? f32x4{0.f, 0.f, 0.f, 0.f} : gx[g];
                gx[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 2 ? xl[ks] : xh[ks], p == 1 ? wl[g][ks] : wh[g][ks], c0, 0, 0, 0);
            }
        };
        float tc[4];
        // cell r of this lane, in place in the accumulators: acc[g][r] -> 2^a -> 1 + 2^(a+b) -> gate value
        auto cA = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[0][r] = __builtin_amdgcn_exp2f(acc[0][r]); acc[1][r] = __builtin_amdgcn_exp2f(acc[1][r]); };
        auto cB = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[2][r] = __builtin_amdgcn_exp2f(acc[2][r]); acc[3][r] = __builtin_amdgcn_exp2f(acc[3][r]); };
        auto cC = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g][r] = __builtin_fmaf(acc[g][r], eb[g], 1.0f);
        };
        auto cD = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[0][r] = __builtin_amdgcn_rcpf(acc[0][r]); acc[1][r] = __builtin_amdgcn_rcpf(acc[1][r]); };
        auto cE = [&](auto r_) { constexpr int r = decltype(r_)::value; acc[2][r] = __builtin_amdgcn_rcpf(acc[2][r]); acc[3][r] = __builtin_amdgcn_rcpf(acc[3][r]); };
        auto cF = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const float g2 = __builtin_fmaf(2.0f * XP_K2, acc[2][r], -XP_K2);          // -2 log2e * tanh(g)
            creg[r] = __builtin_fmaf(acc[1][r], creg[r], acc[0][r] * g2);              // scaled cell state
        };
        auto cG = [&](auto r_) { constexpr int r = decltype(r_)::value; tc[r] = 1.0f + __builtin_amdgcn_exp2f(creg[r]); };
        auto cH = [&](auto r_) { constexpr int r = decltype(r_)::value; tc[r] = __builtin_amdgcn_rcpf(tc[r]); };
        auto cI = [&](auto r_) {
            constexpr int r = decltype(r_)::value;
            const float hv = acc[3][r] * __builtin_fmaf(2.0f, tc[r], -1.0f);
            const _Float16 th = (_Float16)hv;
            const _Float16 tl = (_Float16)(hv - (float)th);
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
        xp_zip<30>(c_mfma, c_ops);
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

int launch_intra_xp(const float* x, const void* w_pk, const float* b_sum, const void* wlin_pk, const float* blin, float* out,
                    int nseq, int nstep, int sdiv, int so, int si, int ps, int dir, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(k_intra_xp, dim3((nseq + 15) / 16), dim3(256), 0, st, x, (const _Float16*)w_pk, b_sum,
                       (const _Float16*)wlin_pk, blin, out, nseq, nstep, sdiv, so, si, ps, dir, accumulate);
    return check_launch();
}

}  // namespace lh
