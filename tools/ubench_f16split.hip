// (1) does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs?  (2) does VALU work overlap with it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define N_ITER 4096

__global__ void k_subnormal(float* out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}

template <int VALU_PER_MFMA, bool TRANS>
__global__ __launch_bounds__(256, 1) void k_same_wave(float* out, float av, float bv) {
    f32x16 acc[4] = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = av + i + threadIdx.x;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU_PER_MFMA; ++q) {
                if (TRANS) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
                else v[q & 7] = fmaf(v[q & 7], av, bv);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    float h[2];
    const float tests[][2] = {{9.5367431640625e-07f, 1.0f}, {3.0e-5f, 1.0f}, {1.0f, 5.9604644775390625e-08f}, {1.0e-3f, 1.0e-3f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, out, t[0], t[1]);
        (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
        printf("a=%g (as f16 %g) b=%g  -> D=%.9g  expected %.9g\n", t[0], h[1], t[1], h[0], 16.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
    }
#define RUN(name, kern) { float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 0.5f, 0.25f); }); \
    printf("%-34s %8.3f ms  cycles/MFMA/SIMD @2.4GHz = %.1f\n", name, ms, ms * 1e-3 * 2.4e9 / (N_ITER * 4)); }
    RUN("f16 mfma, 0 valu", (k_same_wave<0, false>));
    RUN("f16 mfma, 2 fma/mfma", (k_same_wave<2, false>));
    RUN("f16 mfma, 4 fma/mfma", (k_same_wave<4, false>));
    RUN("f16 mfma, 8 fma/mfma", (k_same_wave<8, false>));
    RUN("f16 mfma, 2 exp/mfma", (k_same_wave<2, true>));
    RUN("f16 mfma, 4 exp/mfma", (k_same_wave<4, true>));
    return 0;
}
