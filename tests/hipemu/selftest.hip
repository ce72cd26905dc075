// hipemu self-test: barrier + shuffle reduction + MFMA 16x16x4 GEMM against a scalar reference.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_reduce(const float* x, float* out, int n) {
    __shared__ float part[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[blockIdx.x * n + i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// C[16 x 16] = A[16 x K] * B[K x 16], one wave
__global__ void k_mfma(const float* A, const float* B, float* C, int K) {
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        float a = A[(l & 15) * K + k0 + (l >> 4)];
        float b = B[(k0 + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

extern "C" int selftest() {
    int bad = 0;
    {
        const int n = 1000, nb = 3;
        std::vector<float> x(n * nb), out(nb);
        for (int i = 0; i < n * nb; ++i) x[i] = (float)((i * 7) % 13) - 6.f;
        hipLaunchKernelGGL(k_reduce, dim3(nb), dim3(256), 0, 0, x.data(), out.data(), n);
        for (int b = 0; b < nb; ++b) { float s = 0; for (int i = 0; i < n; ++i) s += x[b * n + i]; if (fabsf(s - out[b]) > 1e-3f) bad++; }
    }
    {
        const int K = 32;
        std::vector<float> A(16 * K), B(K * 16), C(256);
        for (int i = 0; i < 16 * K; ++i) { A[i] = (float)((i * 5) % 11) - 5.f; B[i] = (float)((i * 3) % 7) - 3.f; }
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, A.data(), B.data(), C.data(), K);
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < K; ++k) s += A[i * K + k] * B[k * 16 + j]; if (fabsf(s - C[i * 16 + j]) > 1e-3f) bad++; }
    }
    return bad;
}
