// The single-sequence LSTM step as a quad-lane 256 x 64 mat-vec (round 3, profiles/r03k_stream_step.txt): shared by the separator's
// streaming / batch-1 kernels (lh_stream.hip: k_intra_stream, k_inter_matvec) and the embedder's small-batch inter-axis recurrence
// (lh_embed.hip: k_emb_inter_mv).  One wave per SIMD (256 threads), four lanes per hidden unit: lane (unit u, slice s) keeps the
// 16-wide k slice s of the unit's four gate rows of W_hh (64 fp32 registers), reads its 16 values of h_{t-1} from LDS and does the
// 64 FMAs as 32 packed ones; the four partial sums per gate are reduce-scattered over the quad with DPP quad permutes, so every
// lane evaluates ONE gate non-linearity.
#pragma once
#include "lh_split.h"

namespace lh {

constexpr int IS_GP = 4 * H;               // 256 gate columns, column 4 u + g = gate g of hidden unit u (weights.py)
constexpr int IS_NT = 512;                 // threads of the staging / GEMM phases
constexpr int IS_NW = IS_NT / 64;
constexpr int IS_NR = 256;                 // threads of the recurrence (hidden unit x k slice): waves 0..3, one per SIMD; the
                                           // other four only keep the step barrier company

typedef float f32x2 __attribute__((ext_vector_type(2)));
// value of lane (l ^ 1) / (l ^ 2) of the same quad: DPP quad_perm [1,0,3,2] / [2,3,0,1]
template <int CTRL>
__device__ __forceinline__ float quad_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int QP_XOR1 = 0xB1, QP_XOR2 = 0x4E, QP_XOR3 = 0x1B;

// exponent factor of gate g (PyTorch order i, f, g, o): sigma(x) = 1 / (1 + 2^(-log2e x)), tanh(x) = 2 / (1 + 2^(-2 log2e x)) - 1
__device__ __forceinline__ float quad_gate_scale(int g) { return g == 2 ? -2.0f * LOG2E : -LOG2E; }
constexpr float QS_K2 = -2.0f * LOG2E;     // the cell state is carried times this factor: tanh(c) needs no multiply

// Slot j of lane (u, s) holds the row of gate j ^ s of unit u (row 4 u + (j ^ s) of W_hh [256][64]), columns [16 s, 16 s + 16),
// times the gate's exponent factor: with the gates XOR-rotated by the slice index, lane s finds the partial sums of ITS gate
// in slot x of lane s ^ x — the reduce-scatter is three DPP adds and no selects.
__device__ __forceinline__ void quad_load_w(const float* __restrict__ whh, int u, int s, f32x2 (&w)[4][8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int g = j ^ s;
        const float km = quad_gate_scale(g);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const float4 v = *reinterpret_cast<const float4*>(whh + (long)(4 * u + g) * H + 16 * s + 4 * k4);
            w[j][2 * k4] = f32x2{km * v.x, km * v.y};
            w[j][2 * k4 + 1] = f32x2{km * v.z, km * v.w};
        }
    }
}

// One LSTM step of hidden unit u on its quad of lanes.  hs16 = this lane's 16 values of h_{t-1} (LDS), gxs = the input half
// (+ bias) of gate s of the unit TIMES quad_gate_scale(s).  Returns h_t in the lane with s == 1, which is also where `cs` =
// QS_K2 * cell state lives (the other three lanes run the same instructions on don't-care values that nothing reads; lane 2's
// may overflow to inf / NaN).
// PyTorch gate order i, f, g, o = s.
// QS_PROBE = n: timing probes of the step's parts (WRONG results on purpose; scripts/lab_stream.py): 1 = 4 of the 32 packed
// FMAs, 2 = no transcendentals, 3 = h from registers instead of LDS, 4 = no barrier, 5 = no DPP exchanges
#if !defined(QS_PROBE)
#define QS_PROBE 0
#endif
#if QS_PROBE == 5
#define quad_perm quad_perm_off
template <int CTRL>
__device__ __forceinline__ float quad_perm_off(float v) { return v; }
#endif
#if QS_PROBE == 4
#define QS_SYNC() do { } while (0)
#else
#define QS_SYNC() __syncthreads()
#endif
__device__ __forceinline__ float quad_step(const f32x2 (&w)[4][8], const float* hs16, float gxs, float& cs, int s) {
    f32x2 h2[8];
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) {
#if QS_PROBE == 3
        const float4 v = make_float4(cs, gxs, cs, gxs);
#else
        const float4 v = *reinterpret_cast<const float4*>(hs16 + 4 * k4);
#endif
        h2[2 * k4] = f32x2{v.x, v.y};
        h2[2 * k4 + 1] = f32x2{v.z, v.w};
    }
#if defined(__AMDGCN__)
    // all of h into registers first, ONE wait, then the FMAs: left to itself the scheduler interleaves read / wait / FMAs
    // and exposes the LDS latency every time
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    f32x2 a2[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
    for (int kk = 0; kk < (QS_PROBE == 1 ? 1 : 8); ++kk)
#pragma unroll
        for (int g = 0; g < 4; ++g) a2[g] = __builtin_elementwise_fma(w[g][kk], h2[kk], a2[g]);
    // horizontal sums; one half made opaque: the compiler otherwise forms v_pk_add_f32 with the halves of src1 crossed
    // (op_sel:[0,1] — the unsafe form, build.py)
    auto hsum = [](f32x2 v) {
        float lo = v[0], hi = v[1];
#if defined(__AMDGCN__)
        asm("" : "+v"(hi));
#endif
        return lo + hi;
    };
    const float a0 = hsum(a2[0]), a1 = hsum(a2[1]), a2s = hsum(a2[2]), a3 = hsum(a2[3]);
    // reduce-scatter over the quad: slot x of lane s ^ x is a partial sum of gate s
    const float pre = ((a0 + quad_perm<QP_XOR1>(a1)) + (quad_perm<QP_XOR2>(a2s) + quad_perm<QP_XOR3>(a3))) + gxs;
    // lane 2 (g): K2 tanh = K2 (2 r - 1); the others: sigma = r
    const float ma = s == 2 ? 2.0f * QS_K2 : 1.0f, aa = s == 2 ? -QS_K2 : 0.0f;
#if QS_PROBE == 2
    const float val = fmaf(fmaf(pre, 0.25f, 0.5f), ma, aa);
#else
    const float val = fmaf(__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(pre)), ma, aa);
#endif
    const float vb = quad_perm<QP_XOR2>(val);           // lane 0 (i): K2 tanh(g);  lane 1 (f): sigma(o)
    const float ig = quad_perm<QP_XOR1>(val * vb);      // lane 1: K2 sigma(i) tanh(g)
    cs = fmaf(val, cs, ig);                             // lane 1: K2 c' = sigma(f) K2 c + K2 sigma(i) tanh(g)
#if QS_PROBE == 2
    return vb * (0.5f * cs);
#else
    return vb * fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cs)), -1.0f);      // lane 1: h' = sigma(o) tanh(c')
#endif
}


}  // namespace lh
