// Univariate LSTM-DQN (I = 2, H = 128) with the recurrent product h.U evaluated as a 3-term
// error-compensated f16 split on the 16x-faster half-precision matrix pipe:
//
//     U = U_hi + U_lo,  h = h_hi + h_lo   (each part an f16; hi = round(x), lo = round(x - hi))
//     h.U  ~=  h_hi.U_hi + h_hi.U_lo + h_lo.U_hi          (fp32 accumulate, products exact)
//
// The dropped h_lo.U_lo term is 2^-22 relative and gfx950's v_mfma_f32_32x32x16_f16 honours f16
// subnormal inputs (tools/ubench_f16split.hip), so no scaling is needed; measured against the fp64
// oracle the Q-values are as close as the exact-fp32 kernel's (tools/emulate_f16split.py: 2.0e-7
// vs 1.7e-7 relative on the reference's A1 checkpoint).  3 MFMAs of 32 cycles replace 8 fp32
// MFMAs of 64 cycles per 16 hidden units: 5.3x fewer matrix cycles.  The input k-step
// ([value, flag].Wx) and the bias stay exact fp32 (one v_mfma_f32_32x32x2_f32 per tile).
//
// Structure as lstm_h128i2_kernel: 4 waves = 64 windows in two groups of 32; wave w owns gate
// tiles 4w..4w+3; U_hi/U_lo fragments are register-stationary (AGPR-pinned) for H128F_UREG tiles,
// the rest in LDS; h_t crosses waves through LDS as f16 hi/lo in B-fragment order.
//
// f16 fragment maps (v_mfma_f32_32x32x16_f16): lane l holds A[row = l&31][k = 8*(l>>5)+i] and
// B[k = 8*(l>>5)+i][col = l&31], i = 0..7.  k-step s, lane half hb covers the 8 hidden units of
// group g = 2s+hb, in the order i = 4*half' + q  <->  unit 8g + 2q + half', which is exactly how the
// cell update leaves them (a lane of half' holds units 8g+2q+half', q = 0..3 of gate tile g), so a
// producer lane writes 4 consecutive f16 (one ds_write_b64) and a consumer lane reads one
// ds_read_b128 per fragment.
#pragma once
#include "kernels.hip.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef H128F_UREG
#define H128F_UREG 3
#endif

__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// 128-bit fragment pinned into four accumulator registers (see pin_agpr)
__device__ __forceinline__ f16x8 pin_agpr128(u32x4 x) {
    u32x4 r;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r[0]) : "v"(x[0]));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r[1]) : "v"(x[1]));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r[2]) : "v"(x[2]));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r[3]) : "v"(x[3]));
    return __builtin_bit_cast(f16x8, r);
}

struct LstmF16Args {
    LstmArgs l;
    const u32x4* uhi;     // packed [16 tiles][8 k-steps][64 lanes] x 8 f16
    const u32x4* ulo;
};

__device__ __forceinline__ void h128f_gemm(f32x16 (&acc)[4], const f16x8 (&Uhi)[H128F_UREG][8],
                                           const f16x8 (&Ulo)[H128F_UREG][8], const float (&wx)[4],
                                           const u32x4* __restrict__ ul, const _Float16* __restrict__ hhi,
                                           const _Float16* __restrict__ hlo, const float* __restrict__ bias_l,
                                           float bx, int lane, int half, int j, int w) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        acc[tl] = load_bias16(bias_l, 4 * w + tl, half);
        acc[tl] = mfma32(wx[tl], bx, acc[tl]);
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int off = ((2 * s + half) * 32 + j) * 8;
        const f16x8 bh = *reinterpret_cast<const f16x8*>(hhi + off);
        const f16x8 bl = *reinterpret_cast<const f16x8*>(hlo + off);
#pragma unroll
        for (int tl = 0; tl < H128F_UREG; ++tl) {
            acc[tl] = mfma16(Uhi[tl][s], bh, acc[tl]);
            acc[tl] = mfma16(Ulo[tl][s], bh, acc[tl]);
            acc[tl] = mfma16(Uhi[tl][s], bl, acc[tl]);
        }
#pragma unroll
        for (int tl = H128F_UREG; tl < 4; ++tl) {
            const f16x8 ah = __builtin_bit_cast(f16x8, ul[(((tl - H128F_UREG) * 2 + 0) * 8 + s) * 64 + lane]);
            const f16x8 al = __builtin_bit_cast(f16x8, ul[(((tl - H128F_UREG) * 2 + 1) * 8 + s) * 64 + lane]);
            acc[tl] = mfma16(ah, bh, acc[tl]);
            acc[tl] = mfma16(al, bh, acc[tl]);
            acc[tl] = mfma16(ah, bl, acc[tl]);
        }
    }
}

__device__ __forceinline__ void h128f_pointwise(const f32x16 (&acc)[4], float (&c)[4][4],
                                                _Float16* __restrict__ hhi, _Float16* __restrict__ hlo,
                                                int half, int j, int w) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        f16x4 hi, lo;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float hv = lstm_cell(acc[tl][4 * qd], acc[tl][4 * qd + 1], acc[tl][4 * qd + 2],
                                       acc[tl][4 * qd + 3], 0.0f, c[tl][qd]);
            hi[qd] = (_Float16)hv;
            lo[qd] = (_Float16)(hv - (float)hi[qd]);
        }
        const int off = ((4 * w + tl) * 32 + j) * 8 + half * 4;
        *reinterpret_cast<f16x4*>(hhi + off) = hi;
        *reinterpret_cast<f16x4*>(hlo + off) = lo;
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256, 1) void lstm_h128i2_f16x3_kernel(LstmF16Args fa) {
    const LstmArgs& a = fa.l;
    // separate objects per group so that alias analysis keeps group A's h apart from group B's
    __shared__ __attribute__((aligned(16))) _Float16 s_hiA[16 * 32 * 8];
    __shared__ __attribute__((aligned(16))) _Float16 s_loA[16 * 32 * 8];
    __shared__ __attribute__((aligned(16))) _Float16 s_hiB[16 * 32 * 8];
    __shared__ __attribute__((aligned(16))) _Float16 s_loB[16 * 32 * 8];
    __shared__ __attribute__((aligned(16))) float s_bias[16 * 2 * 16];
    __shared__ __attribute__((aligned(16))) u32x4 s_u[4][(4 - H128F_UREG) * 2 * 8 * 64 + 1];
    __shared__ float s_sv[H128_SV];
    __shared__ float s_sf[H128_SV];
    __shared__ float s_wout[128 * 8];        // Q head, [unit][action]
    __shared__ SegLds s_seg;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    const int cand = blockIdx.y;
    const int W = a.W;
    seg_stage(s_seg, a.seg);
    for (int i = threadIdx.x; i < 128 * a.A; i += 256) s_wout[i] = a.w_out[i];
    // weights to registers once per workgroup; the grid is persistent (see lstm_h128i2_kernel)
    f16x8 Uhi[H128F_UREG][8], Ulo[H128F_UREG][8];
    float wx[4];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) wx[tl] = a.wx[(4 * w + tl) * 64 + lane];
    // all loads of a tile in flight before the first value is pinned (see lstm_h128i2_kernel)
#pragma unroll
    for (int tl = 0; tl < H128F_UREG; ++tl) {
        u32x4 th[8], tlo[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            th[ks] = fa.uhi[((4 * w + tl) * 8 + ks) * 64 + lane];
            tlo[ks] = fa.ulo[((4 * w + tl) * 8 + ks) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            Uhi[tl][ks] = pin_agpr128(th[ks]);
            Ulo[tl][ks] = pin_agpr128(tlo[ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    u32x4* ul = &s_u[w][0];
    for (int tl = H128F_UREG; tl < 4; ++tl)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            ul[(((tl - H128F_UREG) * 2 + 0) * 8 + ks) * 64 + lane] = fa.uhi[((4 * w + tl) * 8 + ks) * 64 + lane];
            ul[(((tl - H128F_UREG) * 2 + 1) * 8 + ks) * 64 + lane] = fa.ulo[((4 * w + tl) * 8 + ks) * 64 + lane];
        }
    for (int i = threadIdx.x; i < 16 * 2 * 16; i += 256) s_bias[i] = a.bias[i];
    const float* bias_l = s_bias;
    __syncthreads();

    for (int tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const SegHit sg = seg_lookup(s_seg, a.seg, tile);
        const int s = sg.s;
        const long long samp0 = sg.samp0, nsamp = sg.nsamp, Ks = sg.Ks;
        const long long k0 = (long long)(tile - sg.tile0) * 64;
        const int count = (int)(Ks - k0 < 64 ? Ks - k0 : 64);
        {
            const bool has_norm = a.seg.in_scale != nullptr;
            const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
            const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
            for (int i = threadIdx.x; i < 64 + W + 1 && i < H128_SV; i += 256) {
                long long n = k0 + i;
                if (n > nsamp - 1) n = nsamp - 1;
                float v = a.x[samp0 + n];
                if (has_norm) v = minmax_apply(v, sc, of);
                s_sv[i] = v;
                s_sf[i] = (a.flag_mode == 2) ? a.flags[samp0 + n] : 0.0f;
            }
        }
        __syncthreads();

        f32x16 accA[4], accB[4];
        float cA[4][4], cB[4][4];
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) { cA[tl][qd] = 0.0f; cB[tl][qd] = 0.0f; }

        auto xin = [&](int g, int t) -> float {
            const int li = g * 32 + j + 1 + t;
            if (half == 0) return s_sv[li];
            if (a.flag_mode == 2) return s_sf[li];
            return synth_flag(a.flag_mode, k0 + g * 32 + j, t, W, cand, nullptr, 0);
        };

        h128_xonly(accA, wx, bias_l, xin(0, 0), half, w);
        h128_xonly(accB, wx, bias_l, xin(1, 0), half, w);
        h128f_pointwise(accA, cA, s_hiA, s_loA, half, j, w);
        __syncthreads();
        for (int t = 1; t < W; ++t) {
            const float bxa = xin(0, t);
            h128f_pointwise(accB, cB, s_hiB, s_loB, half, j, w);
            h128f_gemm(accA, Uhi, Ulo, wx, ul, s_hiA, s_loA, bias_l, bxa, lane, half, j, w);
            __syncthreads();
            const float bxb = xin(1, t);
            h128f_pointwise(accA, cA, s_hiA, s_loA, half, j, w);
            h128f_gemm(accB, Uhi, Ulo, wx, ul, s_hiB, s_loB, bias_l, bxb, lane, half, j, w);
            __syncthreads();
        }
        h128f_pointwise(accB, cB, s_hiB, s_loB, half, j, w);
        __syncthreads();

        // Q head: waves 0/1 -> groups A/B; lane (half, j) sums the 64 units of gate tiles 8*half..8*half+7
        if (w < 2) {
            const _Float16* hh = w == 0 ? s_hiA : s_hiB;
            const _Float16* hl = w == 0 ? s_loA : s_loB;
            const int jw = w * 32 + j;
            ActEpi epi;
            for (int act = 0; act < a.A; ++act) {
                float part = 0.0f;
                for (int g = 8 * half; g < 8 * half + 8; ++g)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int off = (g * 32 + j) * 8 + i;
                        const float hv = (float)hh[off] + (float)hl[off];
                        part = fmaf(hv, s_wout[(8 * g + 2 * (i & 3) + (i >> 2)) * a.A + act], part);
                    }
                part += __shfl_xor(part, 32);
                const float qv = part + a.b_out[act];
                epi.push(act, qv);
                if (half == 0 && jw < count)
                    a.q[((sg.win0 + k0 + jw) * a.S + cand) * a.A + act] = qv;
            }
            if (a.S == 1 && half == 0 && jw < count) epi.store(a, sg.win0 + k0 + jw);
        }
        __syncthreads();        // h and the staged samples are rewritten by the next tile
    }
}
