// Which clock does the chip hold under fp32 MFMA loops of the two shapes, bare and fed from LDS?  (MI355X_MICROARCH.md,
// "DVFS give-back": the clock falls under load, by an amount that depends on the MFMA shape and on the energy per MFMA.)
// One wave per SIMD, every CU, random operands, back-to-back launches for ~2 s per variant; the clock is
// s_memtime / s_memrealtime x 100 MHz around the loop of workgroup 0, the rate is flop / wall time of the last launches.
// build: hipcc -O3 --offload-arch=gfx950 -o build/ubench_mfma_clock tools/ubench_mfma_clock.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_ITER 12000

__device__ unsigned long long g_clk[4];
struct Stamp {
    bool on;
    __device__ Stamp() : on(blockIdx.x == 0 && threadIdx.x == 0) {
        if (on) { g_clk[0] = __builtin_amdgcn_s_memtime(); g_clk[1] = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ ~Stamp() {
        if (on) { g_clk[2] = __builtin_amdgcn_s_memtime(); g_clk[3] = __builtin_amdgcn_s_memrealtime(); }
    }
};

// MODE 0: operands in registers; 1: A operand from LDS (one ds_read_b128 per 4 MFMAs, as the VAE kernels read their
// fragments); 2: as 1 plus one v_fma_f32 per MFMA
template <int MODE>
__global__ __launch_bounds__(256, 1) void k32(const float* __restrict__ in, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i];
    __syncthreads();
    Stamp st;
    const int lane = threadIdx.x & 63;
    f32x16 acc[4] = {};
    float b[4], v[4];
    f32x4 a = *reinterpret_cast<const f32x4*>(in + 4 * threadIdx.x);
    for (int i = 0; i < 4; ++i) { b[i] = in[1024 + 4 * threadIdx.x + i]; v[i] = b[i]; }
    for (int it = 0; it < N_ITER; ++it) {
        if (MODE >= 1) a = *reinterpret_cast<const f32x4*>(lds + ((it & 63) * 256 + lane * 4));
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m], acc[m], 0, 0, 0);
            if (MODE == 2) v[m] = fmaf(v[m], 0.999f, b[m]);
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k16(const float* __restrict__ in, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = in[i];
    __syncthreads();
    Stamp st;
    const int lane = threadIdx.x & 63;
    f32x4 acc[8] = {};
    float b[4], v[4];
    f32x4 a[2];
    a[0] = *reinterpret_cast<const f32x4*>(in + 4 * threadIdx.x);
    a[1] = *reinterpret_cast<const f32x4*>(in + 2048 + 4 * threadIdx.x);
    for (int i = 0; i < 4; ++i) { b[i] = in[1024 + 4 * threadIdx.x + i]; v[i] = b[i]; }
    // the same flop per iteration as k32: 8 MFMAs of half the size
    for (int it = 0; it < N_ITER; ++it) {
        if (MODE >= 1) {
            a[0] = *reinterpret_cast<const f32x4*>(lds + ((it & 31) * 512 + lane * 4));
            a[1] = *reinterpret_cast<const f32x4*>(lds + ((it & 31) * 512 + 256 + lane * 4));
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m >> 2][m & 3], b[m & 3], acc[m], 0, 0, 0);
            if (MODE == 2 && (m & 1)) v[m >> 1] = fmaf(v[m >> 1], 0.999f, b[m >> 1]);
        }
    }
    float s = v[0] + v[1] + v[2] + v[3];
    for (int m = 0; m < 8; ++m) for (int r = 0; r < 4; ++r) s += acc[m][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, const float* d_in, float* d_out, int n_cu) {
    using clk = std::chrono::steady_clock;
    const double flop = (double)n_cu * 4 * N_ITER * 4 * 4096.0;       // per launch
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const auto t0 = clk::now();
    float ms = 0;
    int launches = 0;
    while (std::chrono::duration<double>(clk::now() - t0).count() < 2.0) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(n_cu), dim3(256), 0, 0, d_in, d_out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        launches += 20;
    }
    unsigned long long c[4];
    hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c));
    const double mhz = (double)(c[2] - c[0]) / (double)(c[3] - c[1]) * 100.0;
    const double cyc_per_4096flop = (double)(c[2] - c[0]) / (4.0 * N_ITER);
    printf("%-34s clock %5.0f MHz  %6.1f TFLOP/s  %5.1f cycles per 4096 flop per SIMD  (%d launches, %.3f ms each)\n", name, mhz,
           flop * 20 / (ms * 1e-3) / 1e12, cyc_per_4096flop, launches, ms / 20);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int n_cu = p.multiProcessorCount;
    std::vector<float> h(16384);
    srand(5);
    for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 0.02f;
    float *d_in, *d_out;
    hipMalloc(&d_in, h.size() * 4);
    hipMalloc(&d_out, (size_t)n_cu * 256 * 4);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run("32x32x2  registers", k32<0>, d_in, d_out, n_cu);
    run("16x16x4  registers", k16<0>, d_in, d_out, n_cu);
    run("32x32x2  A from LDS", k32<1>, d_in, d_out, n_cu);
    run("16x16x4  A from LDS", k16<1>, d_in, d_out, n_cu);
    run("32x32x2  A from LDS + VALU", k32<2>, d_in, d_out, n_cu);
    run("16x16x4  A from LDS + VALU", k16<2>, d_in, d_out, n_cu);
    return 0;
}
