// Does v_mfma_f32_16x16x4_f32 accumulate its four k-products in the same order, with the same
// roundings, as two v_mfma_f32_32x32x2_f32 (and as an fmaf chain)?  One wave; compares bits.
// build: hipcc -O2 --offload-arch=gfx950 -o build/ubench_mfma16 tools/ubench_mfma16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [32 rows][K], B [K][32 cols] -> D32 [32][32] via 32x32x2; D16 = top-left [16][16] via 16x16x4
__global__ void k(const float* A, const float* B, int K, float* D32, float* D16) {
    const int l = threadIdx.x;
    f32x16 c = {};
    for (int kk = 0; kk < K; kk += 2)
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * K + kk + (l >> 5)], B[(kk + (l >> 5)) * 32 + (l & 31)], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D32[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
    f32x4 d = {};
    for (int kk = 0; kk < K; kk += 4)
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * K + kk + (l >> 4)], B[(kk + (l >> 4)) * 32 + (l & 15)], d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D16[(4 * (l >> 4) + r) * 16 + (l & 15)] = d[r];
}

int main() {
    const int K = 128;
    std::vector<float> A(32 * K), B(K * 32);
    srand(3);
    for (auto& x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 2.3f;
    for (auto& x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 1.7f;
    float *dA, *dB, *d32, *d16;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d32, 32 * 32 * 4); hipMalloc(&d16, 16 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, K, d32, d16);
    std::vector<float> h32(32 * 32), h16(16 * 16);
    hipMemcpy(h32.data(), d32, h32.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h16.data(), d16, h16.size() * 4, hipMemcpyDeviceToHost);
    int diff_mfma = 0, diff_chain32 = 0, diff_chain16 = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            float ch = 0.0f;
            for (int kk = 0; kk < K; ++kk) ch = __builtin_fmaf(A[r * K + kk], B[kk * 32 + c], ch);
            const float a = h32[r * 32 + c], b = h16[r * 16 + c];
            diff_mfma += memcmp(&a, &b, 4) != 0;
            diff_chain32 += memcmp(&a, &ch, 4) != 0;
            diff_chain16 += memcmp(&b, &ch, 4) != 0;
        }
    printf("of 256 outputs: 16x16x4 != 32x32x2: %d; 32x32x2 != fmaf chain: %d; 16x16x4 != fmaf chain: %d\n",
           diff_mfma, diff_chain32, diff_chain16);
    return 0;
}
