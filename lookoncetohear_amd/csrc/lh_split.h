// Split-precision ("f16x3") building blocks shared by the pointwise, streaming and back-end kernels:
// A-operand LDS images, resident B fragments and the three-MFMA tile product (see lh_pointwise.hip for the rationale).
// Split form everywhere in the separator: v = hi + lo with hi = fp16(v), lo = fp16(v - hi), lo NOT rescaled (the matrix
// core takes fp16 subnormals at full value, profiles/r02a_ubench_issue_model.txt): lo keeps 11 bits while |v| >= 2^-3 and
// an absolute 2^-25 below that, the three partial products go into ONE fp32 accumulator, and a split costs no multiply.
#pragma once
#include "lh_common.h"

namespace lh {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

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
    f16x2_t h01, l01, h23, l23;
    split_pair(v.x, v.y, h01, l01);
    split_pair(v.z, v.w, h23, l23);
    const f16x4 h4 = f16x4{h01[0], h01[1], h23[0], h23[1]}, l4 = f16x4{l01[0], l01[1], l23[0], l23[1]};
    const int idx = a_index<RP>(row, k0);
    *reinterpret_cast<f16x4*>(&ahi[idx]) = h4;
    *reinterpret_cast<f16x4*>(&alo[idx]) = l4;
}

// bias + A[m-tile rows] * W[n-tile]  for K = 32*KS; wh/wl = hi/lo B fragments of this wave's n-tile
template <int RP, int KS>
__device__ __forceinline__ f32x4 mma_tile(const _Float16* ahi, const _Float16* alo, int m, int g4, int l15,
                                          const f16x8 (&wh)[KS], const f16x8 (&wl)[KS], float bias) {
    // two accumulator chains (hi*hi | cross terms) although the un-rescaled split would allow one: shorter dependent
    // MFMA chains for the waves that have nothing else to issue (the difference measured within noise, ~1 %)
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
    // component-wise: a vector add would lower to v_pk_add_f32 (see build.py on packed fp32)
    return f32x4{am[0] + ac[0], am[1] + ac[1], am[2] + ac[2], am[3] + ac[3]};
}

// The same product on ONE accumulator chain (the un-rescaled split allows it): no am + ac adds in the epilogue — 4 vector
// instructions per tile fewer, which is what counts in the VALU-bound frame kernels (two co-resident workgroups cover the
// longer dependent MFMA chain).
template <int RP, int KS>
__device__ __forceinline__ f32x4 mma_tile1(const _Float16* ahi, const _Float16* alo, int m, int g4, int l15,
                                           const f16x8 (&wh)[KS], const f16x8 (&wl)[KS], float bias) {
    f32x4 am = f32x4{bias, bias, bias, bias};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int idx = a_slot<RP>(ks * 4 + g4, m * 16 + l15);
        const f16x8 ah = *reinterpret_cast<const f16x8*>(&ahi[idx]);
        const f16x8 al = *reinterpret_cast<const f16x8*>(&alo[idx]);
        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, wh[ks], am, 0, 0, 0);      // small terms first
        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wl[ks], am, 0, 0, 0);
        am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, wh[ks], am, 0, 0, 0);
    }
    return am;
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
// Range-safe variant for frames of the UN-NORMALISED residual stream (k_qkv_proj_ln): the rows a WAVE stages (rows 4w .. 4w+3
// of every group of 16: each row = 16 lanes of one wave) are multiplied by one power of two taken from their common maximum
// before the split (pow2_scale, lh_common.h); `rinv[row]` receives 1 / scale for the accumulator epilogue (row r of the
// product = rinv[r] * acc).  One wave-level maximum per frame (14 max3 + 4 DPP + 4 readlane, the scale itself on the scalar
// unit) instead of one 16-lane reduction per row: the first form of this function cost the VALU-bound QKV kernel +9 %.
// Precision: 22 bits relative to the largest of the wave's 28 rows, with 2^-37 of it as absolute floor (TE = 12).
constexpr int FR_TE = 12;                  // group maximum scaled into [2^12, 2^13): 3 bits of headroom below 65504
__device__ __forceinline__ void frame_store_scaled(_Float16* ahi, _Float16* alo, float* rinv, int tid,
                                                   const float4 (&stg)[FR_NLD]) {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < FR_NLD; ++i) m = fmaxf(m, absmax4(stg[i]));      // (slots past the frame hold clamped duplicates)
    float s, inv;
    pow2_scale<FR_TE>(wave_max_uniform(m), s, inv);
#pragma unroll
    for (int i = 0; i < FR_NLD; ++i) {
        const int e = tid + 256 * i;
        if (e < NF * 16) {
            store_split4<FR_RP>(ahi, alo, e >> 4, (e & 15) * 4, make_float4(stg[i].x * s, stg[i].y * s, stg[i].z * s, stg[i].w * s));
            if ((e & 15) == 0) rinv[e >> 4] = inv;
        }
    }
}
// rows 97..111 of the image feed only accumulator rows that are dropped, but must hold finite numbers
__device__ __forceinline__ void frame_zero_pad(_Float16* ahi, _Float16* alo, int tid) {
    for (int e = tid; e < (FR_RP - NF) * 16; e += 256)
        store_split4<FR_RP>(ahi, alo, NF + (e >> 4), (e & 15) * 4, make_float4(0.f, 0.f, 0.f, 0.f));
}

}  // namespace lh
