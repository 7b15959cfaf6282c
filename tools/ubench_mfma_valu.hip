// Micro-benchmark: does VALU work overlap with v_mfma_f32_32x32x2_f32 (a) inside one wave,
// (b) across the two waves of a SIMD?  Timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define N_ITER 4096

typedef float f32x2 __attribute__((ext_vector_type(2)));
// packed fp32 (v_pk_fma_f32: two fmas per lane per instruction) next to the MFMA stream
template <int PK_PER_MFMA>
__global__ __launch_bounds__(256, 1) void k_same_wave_pk(float* out, float a, float b) {
    f32x16 acc[4] = {};
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i + threadIdx.x, b + i};
    const f32x2 a2 = {a, a}, b2 = {b, b};
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < PK_PER_MFMA; ++q) v[q & 7] = __builtin_elementwise_fma(v[q & 7], a2, b2);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VALU_PER_MFMA, bool TRANS>
__global__ __launch_bounds__(256, 1) void k_same_wave(float* out, float a, float b) {
    f32x16 acc[4] = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VALU_PER_MFMA; ++q) {
                if (TRANS) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
                else v[q & 7] = fmaf(v[q & 7], a, b);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 512 threads: waves 0-3 MFMA only, waves 4-7 VALU only (same instruction totals as k_same_wave)
template <int VALU_PER_MFMA, bool TRANS>
__global__ __launch_bounds__(512, 2) void k_two_waves(float* out, float a, float b) {
    const int wave = threadIdx.x >> 6;
    float s = 0;
    if (wave < 4) {
        f32x16 acc[4] = {};
        for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
        }
        for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    } else {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
        for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
            for (int q = 0; q < 4 * VALU_PER_MFMA; ++q) {
                if (TRANS) v[q & 7] = __builtin_amdgcn_exp2f(v[q & 7]);
                else v[q & 7] = fmaf(v[q & 7], a, b);
            }
        }
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const double mf = 256.0 * 4 * N_ITER * 4;   // MFMAs per launch
#define RUN(name, kern, threads) { float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 0.5f, 0.25f); }); \
    printf("%-34s %8.3f ms  cycles/MFMA/SIMD @2.4GHz = %.1f\n", name, ms, ms * 1e-3 * 2.4e9 / (N_ITER * 4)); }
    RUN("same wave, 0 valu", (k_same_wave<0, false>), 256);
    RUN("same wave, 4 fma/mfma", (k_same_wave<4, false>), 256);
    RUN("same wave, 8 fma/mfma", (k_same_wave<8, false>), 256);
    RUN("same wave, 12 fma/mfma", (k_same_wave<12, false>), 256);
    RUN("same wave, 4 exp/mfma", (k_same_wave<4, true>), 256);
    RUN("same wave, 4 pk_fma/mfma", (k_same_wave_pk<4>), 256);
    RUN("same wave, 8 pk_fma/mfma", (k_same_wave_pk<8>), 256);
    RUN("two waves, 4 fma/mfma", (k_two_waves<4, false>), 512);
    RUN("two waves, 8 fma/mfma", (k_two_waves<8, false>), 512);
    RUN("two waves, 12 fma/mfma", (k_two_waves<12, false>), 512);
    RUN("two waves, 4 exp/mfma", (k_two_waves<4, true>), 512);
    RUN("two waves, 8 exp/mfma", (k_two_waves<8, true>), 512);
    (void)mf;
    return 0;
}
