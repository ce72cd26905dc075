// Does v_cvt_pk_f16_f32 round like v_cvt_f16_f32 on gfx950?  (test tooling; built on the CPU side, run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
__global__ void k(const float* x, unsigned* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    unsigned pk, s0, s1;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s0) : "v"(a));
    asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s1) : "v"(b));
    const _Float16 c0 = (_Float16)a, c1 = (_Float16)b;          // what the compiler emits for a plain conversion
    unsigned short u0, u1;
    memcpy(&u0, &c0, 2); memcpy(&u1, &c1, 2);
    out[4 * i + 0] = pk;
    out[4 * i + 1] = (s0 & 0xffff) | (s1 << 16);
    out[4 * i + 2] = u0 | ((unsigned)u1 << 16);
    out[4 * i + 3] = 0;
}
int main() {
    const int n = 1 << 20;
    float* h = (float*)malloc(2 * n * sizeof(float));
    srand(1);
    for (int i = 0; i < 2 * n; ++i) {
        const float m = (float)rand() / RAND_MAX * 2.f - 1.f;
        const int e = rand() % 24 - 18;
        h[i] = ldexpf(m, e);
    }
    float* d; unsigned* o;
    hipMalloc(&d, 2 * n * sizeof(float)); hipMalloc(&o, 4 * n * sizeof(unsigned));
    hipMemcpy(d, h, 2 * n * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o, n);
    unsigned* r = (unsigned*)malloc(4 * n * sizeof(unsigned));
    hipMemcpy(r, o, 4 * n * sizeof(unsigned), hipMemcpyDeviceToHost);
    long mis_pk = 0, mis_c = 0; int shown = 0;
    for (int i = 0; i < n; ++i) {
        if (r[4 * i] != r[4 * i + 1]) {
            ++mis_pk;
            if (shown++ < 5) printf("  x = (%.9g, %.9g): v_cvt_pk_f16_f32 -> %08x, 2 x v_cvt_f16_f32 -> %08x\n", h[2 * i], h[2 * i + 1], r[4 * i], r[4 * i + 1]);
        }
        if (r[4 * i + 2] != r[4 * i + 1]) ++mis_c;
    }
    printf("pairs %d: v_cvt_pk_f16_f32 != v_cvt_f16_f32 in %ld pairs; compiler's (_Float16) conversion != v_cvt_f16_f32 in %ld pairs\n", n, mis_pk, mis_c);
    return 0;
}
