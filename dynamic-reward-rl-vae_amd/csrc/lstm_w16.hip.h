// Univariate LSTM-DQN (I = 2, H <= 128) on 16-window tiles: v_mfma_f32_16x16x4_f32.
//
// A call with few windows (one Yahoo series, a replay batch) cannot fill 256 CUs with 32-window tiles,
// and a 32-column MFMA costs the same whether its columns are used or not.  The 16x16x4 form has the
// same flop rate per cycle and half the columns: twice the workgroups, half the work in each, so the
// latency of such a call halves again.  It accumulates its four k-products in order with fmaf
// roundings (tools/ubench_mfma16.hip: bit-identical to two 32x32x2 steps and to an fmaf chain), so a
// window scores to the same bits here as in lstm_h128i2_kernel.
//
// Layout (pack in drl_set_lstm): gate tile T = 16 rows = 4 hidden units x {i,j,f,o}; accumulator
// lane (q = l>>4, col = l&15), register r = row 4q+r = gate r of unit 4T+q of window col — the four
// gates of a cell are the four accumulator registers of one lane.  k-step kk covers units 4kk..4kk+3,
// lane (q, col) supplying unit 4kk+q: exactly the h the lane computed for tile kk, so h_t is written
// to LDS at [tile][lane] and read back at [k-step][lane] with no permutation.
// Wave w owns tiles 8w..8w+7; W16_UREG of them keep their 32 A fragments in registers, the rest in LDS.
#pragma once

#define W16_UREG 6

struct Lstm16Args {
    LstmArgs l;
    const float* u16;      // [32 k-steps][32 tiles][64 lanes]
    const float* wx16;     // [32 tiles][64 lanes]: k = 0 value weight, k = 1 flag weight, k = 2,3 zero
    const float* bias16;   // [32 tiles][4 q][4 gates]
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 1) void lstm_h128i2_w16_kernel(Lstm16Args fa) {
    const LstmArgs& a = fa.l;
    __shared__ __attribute__((aligned(16))) float s_h[2][32 * 64];         // [step parity][k-step][lane]
    __shared__ __attribute__((aligned(16))) float s_bias[32 * 16];
    __shared__ __attribute__((aligned(16))) float s_u[4][(8 - W16_UREG) * 32 * 64 + 4];
    __shared__ float s_sv[H128_SV];
    __shared__ float s_sf[H128_SV];
    __shared__ float s_wout[128 * 8];
    __shared__ SegLds s_seg;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, col = lane & 15;
    const int cand = blockIdx.y;
    const int W = a.W;
    seg_stage(s_seg, a.seg);
    for (int i = threadIdx.x; i < 128 * a.A; i += 256) s_wout[i] = a.w_out[i];
    for (int i = threadIdx.x; i < 32 * 16; i += 256) s_bias[i] = fa.bias16[i];

    float U[W16_UREG][32], wx[8];
#pragma unroll
    for (int tl = 0; tl < 8; ++tl) wx[tl] = fa.wx16[(8 * w + tl) * 64 + lane];
#pragma unroll
    for (int tl = 0; tl < W16_UREG; ++tl) {
        float tmp[32];
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) tmp[kk] = fa.u16[((size_t)kk * 32 + 8 * w + tl) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) U[tl][kk] = pin_agpr(tmp[kk]);
        __builtin_amdgcn_sched_barrier(0);
    }
    float* ul = &s_u[w][0];
    for (int tl = W16_UREG; tl < 8; ++tl) {
#pragma unroll 16
        for (int kk = 0; kk < 32; ++kk)
            ul[((tl - W16_UREG) * 32 + kk) * 64 + lane] = fa.u16[((size_t)kk * 32 + 8 * w + tl) * 64 + lane];
    }
    const float fb = a.forget_bias;
    __syncthreads();

    for (int ti = blockIdx.x; ti < a.n_tiles; ti += gridDim.x) {
        const int tile = a.tile_list ? a.tile_list[ti] : ti;
        const SegHit sg = seg_lookup(s_seg, a.seg, tile);
        const int s = sg.s;
        const long long samp0 = sg.samp0, nsamp = sg.nsamp;
        const long long k0 = (long long)(tile - sg.tile0) * 16;
        const int count = (int)(sg.Ks - k0 < 16 ? sg.Ks - k0 : 16);
        if (a.flag_mode >= 0) {
            const bool has_norm = a.seg.in_scale != nullptr;
            const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
            const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
            for (int i = threadIdx.x; i < 16 + W + 1 && i < H128_SV; i += 256) {
                long long n = k0 + i;
                if (n > nsamp - 1) n = nsamp - 1;
                float v = a.x[samp0 + n];
                if (has_norm) v = minmax_apply(v, sc, of);
                s_sv[i] = v;
                s_sf[i] = (a.flag_mode == 2) ? a.flags[samp0 + n] : 0.0f;
            }
        }
        __syncthreads();

        // B operand of the input k-step: k = 0 the value, k = 1 the flag, k = 2,3 nothing
        auto xin = [&](int t) -> float {
            if (q >= 2) return 0.0f;
            if (a.flag_mode < 0) {
                int jw = col;
                if (jw > count - 1) jw = count - 1;
                return a.x[(samp0 + k0 + jw) * a.win_stride + (long long)t * a.row_stride + q];
            }
            const int li = col + 1 + t;
            if (q == 0) return s_sv[li];
            if (a.flag_mode == 2) return s_sf[li];
            return synth_flag(a.flag_mode, k0 + col, t, W, cand, nullptr, 0);
        };
        auto init = [&](f32x4 (&acc)[8], float bx) {
#pragma unroll
            for (int tl = 0; tl < 8; ++tl) {
                acc[tl] = *reinterpret_cast<const f32x4*>(s_bias + ((8 * w + tl) * 4 + q) * 4);
                acc[tl] = mfma16(wx[tl], bx, acc[tl]);
            }
        };
        // h_t goes to the buffer h_{t-1} is NOT read from, so a step needs one barrier (h_t complete), not two
        auto pointwise = [&](const f32x4 (&acc)[8], float (&c)[8], float* __restrict__ hw) {
#pragma unroll
            for (int tl = 0; tl < 8; ++tl) {
                const float hv = lstm_cell(acc[tl][0], acc[tl][1], acc[tl][2], acc[tl][3], fb, c[tl]);
                hw[(8 * w + tl) * 64 + lane] = hv;
            }
        };

        f32x4 acc[8];
        float c[8];
#pragma unroll
        for (int tl = 0; tl < 8; ++tl) c[tl] = 0.0f;
        init(acc, xin(0));
        pointwise(acc, c, s_h[1]);
        float nx = xin(W > 1 ? 1 : 0);
        __syncthreads();
        for (int t = 1; t < W; ++t) {
            const float bx = nx;
            nx = xin(t + 1 < W ? t + 1 : t);
            init(acc, bx);
            const float* __restrict__ hr = s_h[t & 1];          // h_{t-1}
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                const float b = hr[kk * 64 + lane];
#pragma unroll
                for (int tl = 0; tl < W16_UREG; ++tl) acc[tl] = mfma16(U[tl][kk], b, acc[tl]);
#pragma unroll
                for (int tl = W16_UREG; tl < 8; ++tl) acc[tl] = mfma16(ul[((tl - W16_UREG) * 32 + kk) * 64 + lane], b, acc[tl]);
            }
            pointwise(acc, c, s_h[(t + 1) & 1]);
            __syncthreads();                    // h_t is complete (and every wave has read h_{t-1})
        }
        const float* __restrict__ hfin = s_h[W & 1];            // h of the last step

        // Q head: wave 0, lane (half, window); even and odd units as two chains, as the other kernels do
        if (w == 0 && lane < 32) {
            const int half = lane >> 4;
            ActEpi epi;
            for (int act = 0; act < a.A; ++act) {
                float part = 0.0f;
#pragma unroll 8
                for (int m = 0; m < 64; ++m) {
                    const int unit = 2 * m + half;
                    part = fmaf(hfin[(unit >> 2) * 64 + (unit & 3) * 16 + col], s_wout[unit * a.A + act], part);
                }
                part += __shfl_xor(part, 16);
                const float qv = part + a.b_out[act];
                epi.push(act, qv);
                if (half == 0 && col < count)
                    a.q[((sg.win0 + k0 + col) * a.S + cand) * a.A + act] = qv;
            }
            if (a.S == 1 && half == 0 && col < count) epi.store(a, sg.win0 + k0 + col);
        }
        __syncthreads();
    }
}
