// Energy-per-instruction micro-benchmark (test tooling, built on the CPU side with hipcc, run by scripts/ubench/power_mix.py):
// one instruction class per kernel, every CU busy (256 workgroups x WAVES waves), launched back to back for `seconds`;
// prints instructions per wave executed and the wall time so that the wrapper can turn sampled package power into
// nanojoules per wave-instruction.   usage: power_mix <kind> <waves per workgroup: 4|8> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ITERS = 20000;     // loop trips per launch; each trip = 8 instructions of the class

__global__ void k_mfma16(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.37f + 0.01f * (threadIdx.x & 63) + i); b[i] = (_Float16)(1.1f - 0.02f * i + 0.003f * threadIdx.x); }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += c[i][0] + c[i][3];
    if (s == 1.2345f) out[0] = s;
}
__global__ void k_mfma32(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.37f + 0.01f * (threadIdx.x & 63) + i); b[i] = (_Float16)(1.1f - 0.02f * i + 0.003f * threadIdx.x); }
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) c[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[i & 3], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
    if (s == 1.2345f) out[0] = s;
}
__global__ void k_fma(float* out, int iters) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 0.5f + 0.001f * threadIdx.x + i;
    const float a = 0.999f + 1e-6f * threadIdx.x, b = 1e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 1.2345f) out[0] = s;
}
__global__ void k_exp(float* out, int iters) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = -0.5f - 0.001f * threadIdx.x - 0.1f * i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = -__builtin_amdgcn_exp2f(v[i]);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 1.2345f) out[0] = s;
}
__global__ void k_dsr(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 8 * 4 * 2];
    for (int i = threadIdx.x; i < 64 * 8 * 4 * 2; i += blockDim.x) lds[i] = 1.0f + i;
    __syncthreads();
    f32x4 acc = {0, 0, 0, 0};
    const int base = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(&lds[base + ((i + it) & 7) * 256]);
            acc += t;
        }
    }
    if (acc[0] + acc[3] == 1.2345f) out[0] = acc[1];
}
__global__ void k_mix(float* out, int iters) {       // the recurrent step's mix per 8 MFMAs: 13 fma + 6 exp + 3 ds_read_b128
    __shared__ __attribute__((aligned(16))) float lds[64 * 8 * 4];
    for (int i = threadIdx.x; i < 64 * 8 * 4; i += blockDim.x) lds[i] = 1.0f + i;
    __syncthreads();
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.37f + 0.01f * (threadIdx.x & 63) + i); b[i] = (_Float16)(1.1f - 0.02f * i + 0.003f * threadIdx.x); }
    f32x4 c[8];
    for (int i = 0; i < 8; ++i) c[i] = f32x4{0, 0, 0, 0};
    float v[13], e[6];
    for (int i = 0; i < 13; ++i) v[i] = 0.5f + 0.001f * threadIdx.x + i;
    for (int i = 0; i < 6; ++i) e[i] = -0.5f - 0.001f * threadIdx.x - 0.1f * i;
    f32x4 acc = {0, 0, 0, 0};
    const int base = (threadIdx.x & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[i], 0, 0, 0);
            v[i] = __builtin_fmaf(v[i], 0.999f, 1e-3f);
            if (i < 5) v[8 + i] = __builtin_fmaf(v[8 + i], 0.999f, 1e-3f);
            if (i < 6) e[i] = -__builtin_amdgcn_exp2f(e[i]);
            if (i < 3) acc += *reinterpret_cast<const f32x4*>(&lds[base + ((i + it) & 7) * 256]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = acc[0];
    for (int i = 0; i < 8; ++i) s += c[i][0];
    for (int i = 0; i < 13; ++i) s += v[i];
    for (int i = 0; i < 6; ++i) s += e[i];
    if (s == 1.2345f) out[0] = s;
}
__global__ void k_idle(float* out, int iters) {
    for (int it = 0; it < iters; ++it) __builtin_amdgcn_s_sleep(8);
    if (iters < 0) out[0] = 1;
}

int main(int argc, char** argv) {
    const char* kind = argc > 1 ? argv[1] : "mfma16";
    const int waves = argc > 2 ? atoi(argv[2]) : 4;
    const double secs = argc > 3 ? atof(argv[3]) : 2.0;
    float* d;
    (void)hipMalloc(&d, 4096);
    void (*k)(float*, int) = nullptr;
    if (!strcmp(kind, "mfma16")) k = k_mfma16;
    else if (!strcmp(kind, "mfma32")) k = k_mfma32;
    else if (!strcmp(kind, "fma")) k = k_fma;
    else if (!strcmp(kind, "exp")) k = k_exp;
    else if (!strcmp(kind, "dsr")) k = k_dsr;
    else if (!strcmp(kind, "mix")) k = k_mix;
    else if (!strcmp(kind, "idle")) k = k_idle;
    else { fprintf(stderr, "unknown kind %s\n", kind); return 2; }
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, 100);
    (void)hipDeviceSynchronize();
    long launches = 0;
    auto t0 = std::chrono::steady_clock::now();
    double el = 0;
    while (el < secs) {
        for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, d, ITERS);
        (void)hipDeviceSynchronize();
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // wave-instructions of the class per second, whole chip
    const double instr = (double)launches * ITERS * 8 * 256 * waves;
    printf("%s waves/wg %d  seconds %.3f  launches %ld  wave-instr/s %.4e  (class instructions per wave per trip: 8)\n", kind, waves, el, launches, instr / el);
    return 0;
}
