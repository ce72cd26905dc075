// TEST INFRASTRUCTURE: controls for the emulator's LDS race detector (tests/hipemu/include/hip/hip_runtime.h, HIPEMU_RACE).
//   fine      two waves exchange through LDS with a barrier in between            -> no report
//   same_wave lanes of ONE wave exchange without a barrier (lockstep, in order)    -> no report
//   racy      two waves exchange without a barrier                                 -> reports
//   war       a buffer is re-written by another wave without a barrier after reads -> reports
#include <hip/hip_runtime.h>
__global__ void k_fine(int* p) { __shared__ int s[128]; s[threadIdx.x] = threadIdx.x; __syncthreads(); p[threadIdx.x] = s[(threadIdx.x + 64) % 128]; }
__global__ void k_same_wave(int* p) { __shared__ int s[128]; s[threadIdx.x] = threadIdx.x; p[threadIdx.x] = s[threadIdx.x ^ 1]; }
__global__ void k_racy(int* p) { __shared__ int s[128]; s[threadIdx.x] = threadIdx.x; p[threadIdx.x] = s[(threadIdx.x + 64) % 128]; }
__global__ void k_war(int* p) {
    __shared__ int s[128];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    p[threadIdx.x] = s[(threadIdx.x + 64) % 128];
    s[threadIdx.x] = 0;                              // missing barrier: the other wave may still be reading this slot
}
extern "C" unsigned long hipemu_race_reports();
// counts[0..3] = new reports of fine / same_wave / racy / war; p = 128 ints of scratch
extern "C" int race_controls(int* p, unsigned long* counts) {
    unsigned long c0 = hipemu_race_reports();
    hipLaunchKernelGGL(k_fine, dim3(2), dim3(128), 0, 0, p);
    unsigned long c1 = hipemu_race_reports();
    hipLaunchKernelGGL(k_same_wave, dim3(1), dim3(128), 0, 0, p);
    unsigned long c2 = hipemu_race_reports();
    hipLaunchKernelGGL(k_racy, dim3(1), dim3(128), 0, 0, p);
    unsigned long c3 = hipemu_race_reports();
    hipLaunchKernelGGL(k_war, dim3(1), dim3(128), 0, 0, p);
    unsigned long c4 = hipemu_race_reports();
    counts[0] = c1 - c0; counts[1] = c2 - c1; counts[2] = c3 - c2; counts[3] = c4 - c3;
    return 0;
}
