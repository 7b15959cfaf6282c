// Micro-benchmark: issue-to-issue latency of a DEPENDENT chain of v_fmac_f32, plain and with a DPP
// row_newbcast operand (the inner loop of lstm_gemv_kernel / rollout_seq_kernel), one and two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
template <bool DPP>
__global__ void k_chain(float* out, const float* in) {
    float u[16];
    for (int i = 0; i < 16; ++i) u[i] = in[i * 64 + (threadIdx.x & 63)];
    float h = in[threadIdx.x & 15], z = 0.0f;
    for (int it = 0; it < N_ITER; ++it) {
        if (DPP) {
            asm volatile(
                "v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                : "+v"(z) : "v"(h), "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]));
        } else {
            asm volatile(
                "v_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %1, %3\n\tv_fmac_f32 %0, %1, %4\n\tv_fmac_f32 %0, %1, %5\n\t"
                "v_fmac_f32 %0, %1, %6\n\tv_fmac_f32 %0, %1, %7\n\tv_fmac_f32 %0, %1, %8\n\tv_fmac_f32 %0, %1, %9"
                : "+v"(z) : "v"(h), "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = z;
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); for (int i = 0; i < 5; ++i) f(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 5;
}
int main() {
    float *out, *in; (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&in, 4096 * 4); (void)hipMemset(in, 0, 4096 * 4);
#define RUN(name, kern, threads) { float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, in); }); \
    printf("%-40s %8.3f ms  cycles per fmac per wave @2.4GHz = %.2f\n", name, ms, ms * 1e-3 * 2.4e9 / (N_ITER * 8)); }
    RUN("plain, 1 wave/SIMD", k_chain<false>, 256);
    RUN("plain, 2 waves/SIMD", k_chain<false>, 512);
    RUN("dpp row_newbcast, 1 wave/SIMD", k_chain<true>, 256);
    RUN("dpp row_newbcast, 2 waves/SIMD", k_chain<true>, 512);
    RUN("dpp row_newbcast, 4 waves/SIMD", k_chain<true>, 1024);
    return 0;
}
