// v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 rows x 4 columns, k = 1): (1) is a chain of them bit-identical to
// an fmaf chain (and so to the 32x32x2 / 16x16x4 kernels)?  (2) issue rate with 4 independent accumulators,
// one wave per SIMD.
// build: hipcc -O2 --offload-arch=gfx950 -o build/ubench_mfma4 tools/ubench_mfma4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [64 rows][K], B [K][4 cols] -> D [64][4]: lane l = (block l>>2, j = l&3); A operand row 4*block + (l&3)
__global__ void k_num(const float* A, const float* B, int K, float* D) {
    const int l = threadIdx.x;
    f32x4 d = {};
    for (int kk = 0; kk < K; ++kk)
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l * K + kk], B[kk * 4 + (l & 3)], d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 2) + r) * 4 + (l & 3)] = d[r];
}
#define N_ITER 4096
__global__ __launch_bounds__(256, 1) void k_rate(float* out, float a, float b) {
    f32x4 acc[4] = {};
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256, 1) void k_rate2(float* out, float a, float b) {
    f32x4 acc[2] = {};
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[m & 1], 0, 0, 0);
    }
    float s = 0;
    for (int m = 0; m < 2; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int K = 128;
    std::vector<float> A(64 * K), B(K * 4), D(64 * 4);
    srand(3);
    for (auto& x : A) x = (rand() / (float)RAND_MAX - 0.5f) * 2.3f;
    for (auto& x : B) x = (rand() / (float)RAND_MAX - 0.5f) * 1.7f;
    float *dA, *dB, *dD, *out;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dD, D.size() * 4);
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_num, dim3(1), dim3(64), 0, 0, dA, dB, K, dD);
    (void)hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int diff = 0;
    for (int r = 0; r < 64; ++r)
        for (int c = 0; c < 4; ++c) {
            float ch = 0.0f;
            for (int kk = 0; kk < K; ++kk) ch = __builtin_fmaf(A[r * K + kk], B[kk * 4 + c], ch);
            diff += memcmp(&D[r * 4 + c], &ch, 4) != 0;
        }
    printf("of 256 outputs: 4x4x1 chain != fmaf chain: %d\n", diff);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int v = 0; v < 2; ++v) {
        auto f = [&] { if (v == 0) hipLaunchKernelGGL(k_rate, dim3(256), dim3(256), 0, 0, out, 0.5f, 0.25f);
                       else hipLaunchKernelGGL(k_rate2, dim3(256), dim3(256), 0, 0, out, 0.5f, 0.25f); };
        f(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%d independent accumulators: %.3f ms, cycles per MFMA per SIMD @2.4GHz = %.2f\n", v == 0 ? 4 : 2, ms, ms * 1e-3 * 2.4e9 / (N_ITER * 4));
    }
    return 0;
}
