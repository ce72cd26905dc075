// Shared compile-time shape constants and device helpers for the gfx950 kernels.
// Shapes are those of the reference configs/tsh.json: the kernels are specialised on them.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>
#include "../../include/lookonce_hip.h"

namespace lh {

constexpr int NFFT = 192;        // stft_chunk_size + stft_pad_size (net.py:32)
constexpr int HOP = 128;         // stft_chunk_size
constexpr int NF = 97;           // n_fft/2 + 1 frequency bins
constexpr int NK = 2 * NF;       // filterbank rows (re | im)
constexpr int NMIC = 2;
constexpr int NSRC = 2;
constexpr int C = 64;            // emb_dim
constexpr int H = 64;            // lstm_hidden_units
constexpr int NH = 4;            // attention heads
constexpr int E = 6;             // ceil(512/97)
constexpr int VD = 16;           // C / NH
constexpr int DQK = NF * E;      // 582
constexpr int DV = NF * VD;      // 1552
// Q / K / V live in HBM as split-precision fp16 pairs (v = hi + lo, lo un-rescaled; 4 bytes per element like fp32) in the
// order the attention kernel's v_mfma_f32_16x16x32_f16 operands want:
//   q, kx rows: 76 blocks of 8 features, each [hi 8 | lo 8] halves (features 582..607 are zero)
//   vx rows:    388 quads of 4 columns, each [hi 4 | lo 4] halves
constexpr int QKB = 76;          // 8-feature blocks per q / kx row (608 >= 582)
constexpr int DQKP = QKB * 8;    // 608
constexpr int LDQKH = QKB * 16;  // 1216 halves per q / kx row (2432 bytes)
constexpr int LDVH = DV * 2;     // 3104 halves per vx row (6208 bytes)
constexpr int KV_PAD = LH_KV_PAD_ROWS;   // zero rows behind the T+49 rows of kx / vx (tile over-read, never written)
constexpr int WIN = 50;          // local_atten_len
constexpr int HIST = WIN - 1;    // 49 history rows
constexpr int NQKV = NH * E * 2 + NH * VD;   // 112 projection outputs (Q 24 | K 24 | V 64)
constexpr int SPK = 256;         // spk_emb_dim
constexpr float LN_EPS = 1e-5f;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate non-linearities on the hardware transcendental units: v_exp_f32 (2^x) and v_rcp_f32, both ~1 ulp, instead of
// libm expf + IEEE division (~10 VALU instructions each).  Saturation is exact: exp2 -> inf gives rcp -> 0.
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * x));
}
// tanh(x) = 2*sigmoid(2x) - 1; abs error ~1e-7
__device__ __forceinline__ float tanh_f(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.0f * LOG2E * x)) - 1.0f;
}
// One LSTM cell update (PyTorch gate order i,f,g,o): c' = sigma(f) c + sigma(i) tanh(g), h' = sigma(o) tanh(c').
// (A variant over common denominators, 5 v_exp + 2 v_rcp instead of 5 + 5, measured the same on the MI355X:
//  profiles/r01i_lab_lstm_attn_variants.txt — the recurrence is not bound by transcendental issue.)
__device__ __forceinline__ void lstm_cell(float gi, float gf, float gg, float go, float& c, float& h) {
    const float ig = sigmoid_f(gi), fg = sigmoid_f(gf), g2 = tanh_f(gg), og = sigmoid_f(go);
    const float cc = fg * c + ig * g2;
    c = cc;
    h = og * tanh_f(cc);
}
// The same cell on PRE-SCALED gate pre-activations: the packed weights / biases of the split-precision recurrent kernels
// carry the exponent scale (rows i, f, o times -log2 e, rows g times -2 log2 e: weights.py GATE_PRESCALE), so
// sigma = rcp(1 + exp2(a)) and tanh = 2 rcp(1 + exp2(a)) - 1 need no multiply on the way in (16 VALU per step).
__device__ __forceinline__ void lstm_cell_pre(float ai, float af, float ag, float ao, float& c, float& h) {
#if defined(LH_PROBE_NOTRANS)      // timing probe only (wrong results): no transcendentals at all
    {
        const float ig = 0.5f + 0.25f * ai, fg = 0.5f + 0.25f * af, g2 = 0.5f * ag, og = 0.5f + 0.25f * ao;
        const float cc0 = fg * c + ig * g2;
        c = cc0;
        h = og * (0.5f * cc0);
        return;
    }
#elif defined(LH_PROBE_NOTANHC)    // timing probe only: h = o * c (two transcendentals fewer)
    {
        const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ai));
        const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(af));
        const float g2 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ag)) - 1.0f;
        const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ao));
        const float cc0 = fg * c + ig * g2;
        c = cc0;
        h = og * cc0;
        return;
    }
#endif
    const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ai));
    const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(af));
    const float g2 = 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ag)) - 1.0f;
    const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ao));
    const float cc = fg * c + ig * g2;
    c = cc;
    h = og * tanh_f(cc);
}
// Wave index of a thread as a SCALAR.  `tid >> 6` is wave-uniform, but the compiler's uniformity analysis does not know the
// wavefront size: every condition and address derived from it counted as divergent — exec-masked branches (s_and_saveexec +
// s_cbranch) around whole MFMA groups "if (wave + 4 < NQKV / 16)", per-lane address arithmetic for per-wave offsets (round 5,
// found by reading the ISA of k_gemm_nt: 137 s_and_saveexec for 48 MFMAs; k_local_attn<3,2,40>: 279).  v_readfirstlane
// makes it an SGPR: scalar branches, scalar address arithmetic, fewer VGPRs.
__device__ __forceinline__ int wave_id(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }

__device__ __forceinline__ float prelu_f(float x, float a) { return x >= 0.0f ? x : a * x; }
// PReLU in two instructions for a slope known to be <= 1 / > 1 (a wave-uniform property of the layer, tested once per
// kernel): for a <= 1 the two branches x and a*x never cross the wrong way — x >= 0: a*x <= x; x < 0: a*x >= x — so
// prelu(x) = max(x, a*x); for a > 1 it is min(x, a*x).  The compare + select form above costs three (frame kernels are
// VALU-bound).  NaN: max / min return the non-NaN operand, but x = NaN makes both operands NaN.
template <bool SLOPE_LE_1>
__device__ __forceinline__ float prelu_mm(float x, float a) {
    return SLOPE_LE_1 ? __builtin_fmaxf(x, a * x) : __builtin_fminf(x, a * x);
}

// sum over the 64 lanes of a wave (every lane gets the total): 4 DPP row steps + 2 cross-row shuffles
__device__ __forceinline__ float wave_sum(float v);
// v + (v rotated right by N lanes inside each aligned row of 16 lanes): one VALU op with a DPP modifier
// (row_ror:N = dpp_ctrl 0x120+N) instead of a ds_bpermute round trip through the LDS crossbar.
template <int N>
__device__ __forceinline__ float row_ror_add(float v) {
    const int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false);
    return v + __builtin_bit_cast(float, r);
}
// value of the lane N positions to the left (cyclic) inside each aligned row of 16 lanes; N = 8 swaps lane l with l ^ 8
template <int N>
__device__ __forceinline__ float row_ror_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
// sum over aligned groups of 16 lanes (every lane of the group gets the total)
__device__ __forceinline__ float group16_sum(float v) {
    v = row_ror_add<8>(v);
    v = row_ror_add<4>(v);
    v = row_ror_add<2>(v);
    return row_ror_add<1>(v);
}

// max of NON-NEGATIVE floats over aligned groups of 16 lanes (every lane of the group gets it).  Done on the bit patterns
// (same order as the values for v >= 0; a NaN sorts above everything, which is what the range guard wants): with the DPP
// source defaulting to 0 — the identity of an unsigned max — hipcc folds each step into ONE v_max_u32_dpp; the float
// version stayed v_mov_b32_dpp + v_max_f32 (0 is not fmax's identity), twice the instructions in VALU-bound frame kernels.
template <int N>
__device__ __forceinline__ unsigned row_ror_umax(unsigned v) {
    const unsigned r = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, false);
    return v > r ? v : r;
}
__device__ __forceinline__ float group16_max(float v) {
    unsigned u = __float_as_uint(v);
    u = row_ror_umax<8>(u);
    u = row_ror_umax<4>(u);
    u = row_ror_umax<2>(u);
    return __uint_as_float(row_ror_umax<1>(u));
}
// wave-uniform max of non-negative floats: 4 folded DPP steps, then the four row leaders through v_readlane / s_max (the
// result lives in an SGPR: everything derived from it — pow2_scale — runs on the scalar unit)
__device__ __forceinline__ float wave_max_uniform(float v) {
    const unsigned u = __float_as_uint(group16_max(v));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)u, 0), b = (unsigned)__builtin_amdgcn_readlane((int)u, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)u, 32), d = (unsigned)__builtin_amdgcn_readlane((int)u, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return __uint_as_float(ab > cd ? ab : cd);
}
__device__ __forceinline__ float wave_max(float v) {
    v = group16_max(v);
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
}

// Range-safe split (range contract, include/lookonce_hip.h).  The fp16 hi half of a split overflows at
// 65504 and loses bits below 2^-14, so a kernel that splits UN-NORMALISED data (waveform, residual stream, spectra)
// first multiplies the row / tile by an exact power of two  s = 2^(TE - floor(log2 m))  that brings its maximum
// magnitude m into [2^TE, 2^(TE+1)), and multiplies the fp32 accumulator by 1/s afterwards (both exact; everything
// between is linear or positively homogeneous).  Precision then is ~22 bits relative to the row maximum for ANY finite
// m — the reference's plain-fp32 behaviour — instead of relative to 1.0 with a hard overflow at 65504.
// m = 0 / subnormal and m = inf / NaN clamp to the ends (a non-finite row stays non-finite and is flagged downstream).
template <int TE>
__device__ __forceinline__ void pow2_scale(float maxabs, float& s, float& inv_s) {
    static_assert(TE >= 1 && TE <= 14, "scaled maximum must stay below the fp16 range");
    int e = (int)((__float_as_uint(maxabs) >> 23) & 0xffu);            // biased exponent of the maximum
    e = min(max(e, TE + 1), 254);
    s = __uint_as_float((unsigned)(254 + TE - e) << 23);               // 2^(TE - (e - 127))
    inv_s = __uint_as_float((unsigned)(e - TE) << 23);                 // 2^((e - 127) - TE)
}
__device__ __forceinline__ float absmax4(const float4& v) {        // two v_max3_f32 with |abs| source modifiers
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(fabsf(v.x), fabsf(v.y)), fabsf(v.z)), fabsf(v.w));
}

__device__ __forceinline__ float wave_sum(float v) {
    v = group16_sum(v);
    v += __shfl_xor(v, 16);
    return v + __shfl_xor(v, 32);
}

// Sum `v` over all threads of a 256-thread workgroup; `red` is a >= 4-float LDS scratch.
// Contains two barriers; every thread must call it.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();                       // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Split v = hi + lo (hi = fp16(v), lo = fp16(v - hi)): an error-free transformation — it only works if `hi` is ONE number.
// hipcc folds fptrunc(fmul a, b) into v_fma_mixlo_f16 (the EXACT product rounded once to fp16) whatever -ffp-contract says,
// and it does so per use: handed `x * rstd`, it computed the `hi` it subtracts from the fused form and the `hi` it stores
// from the rounded fp32 product.  In the rare double-rounding cases the two differ by one fp16 ulp and hi + lo is off by
// 2^-11 |v| — seen as 6e-5 errors of whole LSTM outputs against 3e-7 (round 3; which pattern the compiler picks changes
// with every flag, the SLP build of rounds 1-2 happened to be consistent).  The empty asm makes the operand opaque: the
// product is rounded to fp32 once and both conversions start from that register.
// Round 5: on gfx950 the split is written in the three instructions it needs (the compiler's form was four per value:
// v_cvt_f16_f32, v_cvt_f32_f16, v_sub_f32 + its half of a v_cvt_pk_f16_f32 that converts `hi` a second time for the store):
//   hi = v_cvt_f16_f32(v);  d = v_fma_mix_f32(hi as f16, -1.0, v) = v - hi, exact;  lo = v_cvt_f16_f32(d)
// and `hi` is one register by construction, so no opaque-operand trick is needed.  Same bits as before (both conversions
// round to nearest even, the difference is exact in fp32).  The host build (tests/hipemu) keeps the portable form.
__device__ __forceinline__ void split_hl(float v, _Float16& hi, _Float16& lo) {
#if defined(__AMDGCN__)
    float d;
    asm("v_cvt_f16_f32 %0, %1" : "=v"(hi) : "v"(v));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hi), "v"(v));
    asm("v_cvt_f16_f32 %0, %1" : "=v"(lo) : "v"(d));
#else
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
#endif
}
// Two values at once, results PACKED (element 0 in the low half): v_cvt_pk_f16_f32 (gfx950) + two v_fma_mix_f32 reading
// the halves of the packed register + v_cvt_pk_f16_f32 = 2 instructions per value, packing included (the single form
// plus the compiler's packs: 3; the compiler's own: 4).  Bit-identical to two split_hl.
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float v0, float v1, f16x2_t& hi, f16x2_t& lo) {
#if defined(__AMDGCN__)
    float d0, d1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(v0), "v"(v1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(hi), "v"(v0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(hi), "v"(v1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(d0), "v"(d1));
#else
    _Float16 h0, l0, h1, l1;
    split_hl(v0, h0, l0);
    split_hl(v1, h1, l1);
    hi = f16x2_t{h0, h1};
    lo = f16x2_t{l0, l1};
#endif
}

// "The value of v is decided HERE": an empty volatile asm that claims to rewrite the registers.  Arithmetic on a global
// load's result cannot be hoisted above it — used at the top of an unrolled recurrent step so the consumer of a load
// issued one step earlier (and its s_waitcnt vmcnt) stays behind the step barrier instead of landing right after the load.
__device__ __forceinline__ void pin_here(float4& v) {
#if defined(__AMDGCN__)
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
#else
    (void)v;
#endif
}

// compile-time loops for hand-ordered instruction sequences (lh_recur.hip xp_zip, lh_embed.hip k_emb_rec): f(integral_constant<int, I>) for I < N
template <class F, int... I>
__device__ __forceinline__ void xp_sf(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void xp_for(F&& f) { xp_sf(f, std::make_integer_sequence<int, N>{}); }

inline int check_launch() { return hipGetLastError() == hipSuccess ? LH_OK : LH_ERR_LAUNCH; }

}  // namespace lh
