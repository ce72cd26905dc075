// Micro-benchmark (timing probe, not product code): issue rate of v_pk_fma_f32 against v_fma_f32 for one wave per SIMD, with
// the operand register pairs in the same / in different VGPR bank pairs.  Build: hipcc --offload-arch=gfx950 -O2 pk_rate.hip -o pk_rate
// Each kernel runs NIT x 32 instructions on 8 independent accumulators; time per instruction = (t - t_empty) / (NIT * 32).
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

template <int V>
__global__ void __launch_bounds__(64) k(float* out, int nit) {
    float r = 0.f;
    // registers v[2:3] .. v[40:41]: accumulators at 8.., sources chosen per variant
    asm volatile("v_mov_b32 v2, 1.0\n v_mov_b32 v3, 1.0\n v_mov_b32 v4, 0.5\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 0.25\n v_mov_b32 v7, 0.25\n"
                 "v_mov_b32 v8, 0\n v_mov_b32 v9, 0\n v_mov_b32 v10, 0\n v_mov_b32 v11, 0\n v_mov_b32 v12, 0\n v_mov_b32 v13, 0\n v_mov_b32 v14, 0\n v_mov_b32 v15, 0\n"
                 "v_mov_b32 v16, 0\n v_mov_b32 v17, 0\n v_mov_b32 v18, 0\n v_mov_b32 v19, 0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
                 ::: "v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
    for (int i = 0; i < nit; ++i) {
        if (V == 0) {          // packed, src0 v[2:3] (banks 2,3), src1 v[6:7] (banks 2,3): same bank pair
            asm volatile(REP4("v_pk_fma_f32 v[8:9], v[2:3], v[6:7], v[8:9]\n v_pk_fma_f32 v[10:11], v[2:3], v[6:7], v[10:11]\n v_pk_fma_f32 v[12:13], v[2:3], v[6:7], v[12:13]\n v_pk_fma_f32 v[14:15], v[2:3], v[6:7], v[14:15]\n"
                              "v_pk_fma_f32 v[16:17], v[2:3], v[6:7], v[16:17]\n v_pk_fma_f32 v[18:19], v[2:3], v[6:7], v[18:19]\n v_pk_fma_f32 v[20:21], v[2:3], v[6:7], v[20:21]\n v_pk_fma_f32 v[22:23], v[2:3], v[6:7], v[22:23]\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
        } else if (V == 1) {   // packed, src0 v[2:3] (banks 2,3), src1 v[4:5] (banks 0,1)
            asm volatile(REP4("v_pk_fma_f32 v[8:9], v[2:3], v[4:5], v[8:9]\n v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[10:11]\n v_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[12:13]\n v_pk_fma_f32 v[14:15], v[2:3], v[4:5], v[14:15]\n"
                              "v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[16:17]\n v_pk_fma_f32 v[18:19], v[2:3], v[4:5], v[18:19]\n v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n v_pk_fma_f32 v[22:23], v[2:3], v[4:5], v[22:23]\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
        } else if (V == 2) {   // scalar fma, 32 per iteration
            asm volatile(REP4("v_fma_f32 v8, v2, v6, v8\n v_fma_f32 v9, v2, v6, v9\n v_fma_f32 v10, v2, v6, v10\n v_fma_f32 v11, v2, v6, v11\n"
                              "v_fma_f32 v12, v2, v6, v12\n v_fma_f32 v13, v2, v6, v13\n v_fma_f32 v14, v2, v6, v14\n v_fma_f32 v15, v2, v6, v15\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15");
        } else if (V == 3) {   // v_fmac (VOP2), 32 per iteration
            asm volatile(REP4("v_fmac_f32 v8, v2, v6\n v_fmac_f32 v9, v2, v6\n v_fmac_f32 v10, v2, v6\n v_fmac_f32 v11, v2, v6\n"
                              "v_fmac_f32 v12, v2, v6\n v_fmac_f32 v13, v2, v6\n v_fmac_f32 v14, v2, v6\n v_fmac_f32 v15, v2, v6\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15");
        } else if (V == 4) {   // packed, accumulators alternate bank pairs against the sources: acc even pair index in banks 0,1
            asm volatile(REP4("v_pk_fma_f32 v[8:9], v[2:3], v[4:5], v[8:9]\n v_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[12:13]\n v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[16:17]\n v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n"
                              "v_pk_fma_f32 v[8:9], v[2:3], v[4:5], v[8:9]\n v_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[12:13]\n v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[16:17]\n v_pk_fma_f32 v[20:21], v[2:3], v[4:5], v[20:21]\n") ::: "v8","v9","v12","v13","v16","v17","v20","v21");
        } else if (V == 5) {   // packed mul only (two sources)
            asm volatile(REP4("v_pk_mul_f32 v[8:9], v[2:3], v[4:5]\n v_pk_mul_f32 v[10:11], v[2:3], v[4:5]\n v_pk_mul_f32 v[12:13], v[2:3], v[4:5]\n v_pk_mul_f32 v[14:15], v[2:3], v[4:5]\n"
                              "v_pk_mul_f32 v[16:17], v[2:3], v[4:5]\n v_pk_mul_f32 v[18:19], v[2:3], v[4:5]\n v_pk_mul_f32 v[20:21], v[2:3], v[4:5]\n v_pk_mul_f32 v[22:23], v[2:3], v[4:5]\n") ::: "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
        }
    }
    asm volatile("v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10" : "=v"(r) :: "v8","v9","v10");
    if (r == 12345.f) out[threadIdx.x] = r;
}

template <int V>
float run(float* d, int nit, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, nit);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(64), 0, 0, d, nit);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, 4096);
    const int nit = 200000;
    const char* names[6] = {"v_pk_fma_f32 src0/src1 same bank pair", "v_pk_fma_f32 src0/src1 different bank pairs", "v_fma_f32 (VOP3)", "v_fmac_f32 (VOP2)",
                            "v_pk_fma_f32, accumulators in the bank pair of src1 only", "v_pk_mul_f32"};
    float ms[6];
    for (int rep = 0; rep < 2; ++rep) {
        ms[0] = run<0>(d, nit, 1); ms[1] = run<1>(d, nit, 1); ms[2] = run<2>(d, nit, 1); ms[3] = run<3>(d, nit, 1); ms[4] = run<4>(d, nit, 1); ms[5] = run<5>(d, nit, 1);
    }
    // one wave on one SIMD; clock unknown: report ns per instruction and the ratio to v_fmac
    for (int v = 0; v < 6; ++v)
        printf("%-62s %.3f ns per wave-instruction  (%.2fx v_fmac_f32)\n", names[v], ms[v] * 1e6 / (nit * 32.0), ms[v] / ms[3]);
    return 0;
}
