// Micro-benchmark (not part of the product): can the gfx950 matrix pipe (v_mfma_f32_16x16x32_f16) run under
// vector work (v_fma_f32 / v_exp_f32) of the SAME wave and of ANOTHER wave on the same SIMD?
//   mode 0: MFMA only      mode 1: VALU (fma) only     mode 2: transcendental only
//   mode 3: MFMA + fma interleaved in one wave          mode 4: MFMA + exp interleaved in one wave
//   mode 5: two waves per SIMD, even waves MFMA, odd waves fma      mode 6: same with exp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (lane + i)); b[i] = (_Float16)(0.002f * (lane - i)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.001f * (lane + i);
    auto body = [&](auto DM, auto DV, auto DT) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (DM.value) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
                if (DV.value) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[(u + j) & 7] = __builtin_fmaf(v[(u + j) & 7], 1.0001f, 0.5f);
                }
                if (DT.value) v[u] = __builtin_amdgcn_exp2f(v[u]) * 0.25f;
            }
        }
    };
    using T = std::true_type; using F = std::false_type;
    if (MODE == 0) body(T{}, F{}, F{});
    if (MODE == 1) body(F{}, T{}, F{});
    if (MODE == 2) body(F{}, F{}, T{});
    if (MODE == 3) body(T{}, T{}, F{});
    if (MODE == 4) body(T{}, F{}, T{});
    if (MODE == 5) { if (wave & 4) body(F{}, T{}, F{}); else body(T{}, F{}, F{}); }
    if (MODE == 6) { if (wave & 4) body(F{}, F{}, T{}); else body(T{}, F{}, F{}); }
    if (MODE == 7) { if (wave & 4) { __builtin_amdgcn_s_setprio(0); body(F{}, T{}, F{}); } else { __builtin_amdgcn_s_setprio(3); body(T{}, F{}, F{}); } }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, int threads, float* d) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s threads/WG %3d  %8.3f ms  = %.1f ns per 8-slot iteration\n", name, threads, ms, ms * 1e6 / iters);
}

int main() {
    float* d;
    (void)hipMalloc(&d, 256 * 512 * 4);
    run<0>("MFMA only (8 per iter), 1 wave/SIMD", 256, d);
    run<1>("fma only (32 per iter), 1 wave/SIMD", 256, d);
    run<2>("exp only (8 exp + 8 mul per iter)", 256, d);
    run<3>("MFMA + fma interleaved, same wave", 256, d);
    run<4>("MFMA + exp interleaved, same wave", 256, d);
    run<0>("MFMA only, 2 waves/SIMD", 512, d);
    run<1>("fma only, 2 waves/SIMD", 512, d);
    run<5>("2 waves/SIMD: one MFMA, one fma", 512, d);
    run<6>("2 waves/SIMD: one MFMA, one exp", 512, d);
    run<7>("2 waves/SIMD: MFMA (prio 3), fma (prio 0)", 512, d);
    run<3>("MFMA + fma interleaved, 2 waves/SIMD", 512, d);
    return 0;
}
