// Minimal reproducer attempt for profiles/r03c_packed_fp32_corruption.txt (test tooling): the packed-fp32 variance chain
// of k_proj_ln_res (SLP build), verbatim with its physical registers, in a loop; built as a tiny shared library and driven by
// scripts/ubench/pk_race.py (quiet run vs run next to the LSTM kernel of the product library on a second stream).
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC pk_race.hip -o libpk_race.so  [-DPK_NOP_AFTER_PK=1 ...]
#include <hip/hip_runtime.h>

#ifndef PK_PAD
#define PK_PAD ""                 // e.g. "s_nop 1\n" inserted behind every packed instruction (bisect builds)
#endif
#define PKI(txt) txt "\n" PK_PAD

__global__ void __launch_bounds__(256, 2) k_pk_victim(const float* __restrict__ in, unsigned* __restrict__ out, int iters) {
    __shared__ float lds[256 * 4];
    const int tid = threadIdx.x;
    const float* p = in + ((long)blockIdx.x * 256 + tid) * 16;
    float d[12];
    for (int i = 0; i < 12; ++i) d[i] = p[i];
    const float mean = p[12];
    unsigned acc0 = 0, acc1 = 0;
    for (int it = 0; it < iters; ++it) {
        float r0, r1;
        asm volatile(
            "v_mov_b32 v134, %2\n"
            "v_mov_b32 v182, %3\n v_mov_b32 v183, %4\n v_mov_b32 v180, %5\n v_mov_b32 v181, %6\n"
            "v_mov_b32 v142, %7\n v_mov_b32 v143, %8\n v_mov_b32 v184, %9\n v_mov_b32 v185, %10\n"
            "v_mov_b32 v144, %11\n v_mov_b32 v145, %12\n v_mov_b32 v178, %13\n v_mov_b32 v179, %14\n"
            "v_mov_b32 v141, 0\n"
            "s_nop 4\n"
            "v_mov_b32_e32 v135, v134\n"
            PKI("v_pk_add_f32 v[158:159], v[182:183], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
            PKI("v_pk_add_f32 v[234:235], v[180:181], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
            "v_mul_f32_e32 v140, v159, v159\n"
            PKI("v_pk_add_f32 v[240:241], v[142:143], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
            PKI("v_pk_fma_f32 v[158:159], v[158:159], v[158:159], v[140:141] op_sel_hi:[1,1,0]")
            "v_mul_f32_e32 v140, v235, v235\n"
            PKI("v_pk_add_f32 v[238:239], v[184:185], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
            PKI("v_pk_mul_f32 v[240:241], v[240:241], v[240:241]")
            PKI("v_pk_add_f32 v[158:159], v[140:141], v[158:159] op_sel_hi:[0,1]")
            PKI("v_pk_add_f32 v[236:237], v[144:145], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
            PKI("v_pk_fma_f32 v[238:239], v[238:239], v[238:239], v[240:241]")
            PKI("v_pk_fma_f32 v[158:159], v[234:235], v[234:235], v[158:159]")
            PKI("v_pk_add_f32 v[234:235], v[178:179], v[134:135] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
            PKI("v_pk_fma_f32 v[236:237], v[236:237], v[236:237], v[238:239]")
            "s_nop 0\n"
            PKI("v_pk_fma_f32 v[234:235], v[234:235], v[234:235], v[236:237]")
            "s_nop 0\n"
            PKI("v_pk_add_f32 v[158:159], v[234:235], v[158:159]")
            "s_nop 0\n"
            PKI("v_pk_add_f32 v[158:159], v[158:159], v[234:235] op_sel:[0,1] op_sel_hi:[1,0]")
            "s_nop 4\n"
            "v_mov_b32 %0, v158\n v_mov_b32 %1, v159\n"
            : "=v"(r0), "=v"(r1)
            : "v"(mean), "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]), "v"(d[8]),
              "v"(d[9]), "v"(d[10]), "v"(d[11])
            : "v134", "v135", "v140", "v141", "v142", "v143", "v144", "v145", "v158", "v159", "v178", "v179", "v180", "v181",
              "v182", "v183", "v184", "v185", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241");
        acc0 ^= __float_as_uint(r0) + (unsigned)it;
        acc1 ^= __float_as_uint(r1) + 3u * (unsigned)it;
#if defined(PK_LDS)
        // LDS traffic + barrier between repetitions, like the real kernel's step
        lds[tid * 4 + (it & 3)] = r0;
        __syncthreads();
        d[0] += lds[((tid + 64) & 255) * 4 + (it & 3)] * 0.0f;
#endif
    }
    out[((long)blockIdx.x * 256 + tid) * 2] = acc0;
    out[((long)blockIdx.x * 256 + tid) * 2 + 1] = acc1;
}

extern "C" int pk_victim(const float* in, unsigned* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(k_pk_victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, iters);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
