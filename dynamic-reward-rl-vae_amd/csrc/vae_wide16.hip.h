// K1+K2 for wide windows (D = W*F in the hundreds or thousands: SMD 950, WaDi 3075) on 16-window
// wave tiles: v_mfma_f32_16x16x4_f32.
//
// The first and the last Dense layer are real GEMMs here ([windows, D] x [D, 128] and back).  Their
// weights are streamed slab by slab (one slab = 32 window elements x 128 hidden = 16 KiB, contiguous
// in the packed layout) through a double-buffered LDS ring; the narrow middle layers chain in
// registers.  What differs from the 32-window form it replaces:
//   * a workgroup is 64 consecutive windows of one series (4 waves x 16), and its windows are staged
//     ONCE into LDS as the flat region they overlap in ((63 + W) samples), not once per wave: 44 KB
//     instead of 110 KB for WaDi, so two workgroups (three for SMD) share a CU and every SIMD holds two
//     or three waves of different workgroups — one's barrier, LDS and staging waits run under the
//     other's MFMAs (a single wave per SIMD left 47 % of the matrix pipe idle);
//   * the 16x16x4 form has the same flop rate as 32x32x2 and its k order is the natural element order.
//
// MFMA layouts (v_mfma_f32_16x16x4_f32, lane l: q = l >> 4, col = l & 15):
//   A (1 VGPR)  A[row = col'][k = q]     -> a weight, pre-packed on the host (col' = l & 15 is the row here)
//   B (1 VGPR)  B[k = q][col]            -> element 4*ks + q of window col
//   C/D (4)     D[row = 4q + r][col]     -> register r of lane (q, col)
// so accumulator register r of row tile T is the B operand of the next layer's k-step (T, r), which
// covers units 16T + 4q + r, q = 0..3: activations chain from layer to layer in the register file.
#pragma once
#include <type_traits>

#define VW16_WIN 64          // windows per workgroup
#define VW16_SLAB 4096       // floats per staged weight slab
#define VW16_SIDE 160        // c_hi[32] | c_lo[32] | inv_hi[32] | inv_lo[32] | bias[32]
#ifndef VW16_ABL
#define VW16_ABL 0           // timing-only ablations (wrong results): 1 no scaler transform, 2 no last-layer epilogue,
                             // 3 no global weight loads, 4 no barriers in the slab loops, 5 no LDS stores of staged weights,
                             // 6 one slab of layer 0 only, 7 one slab of the last layer only
#endif
#if VW16_ABL == 4 || VW16_ABL == 11
#define VW16_SYNC() do { } while (0)
#else
#define VW16_SYNC() __syncthreads()
#endif

// VW16_STAMPS (diagnostic builds): s_memtime stamps of layer 0's first slabs, workgroup 0, wave 0
#ifdef VW16_STAMPS
__device__ unsigned long long vw16_dbg[64 * 5];
#define VW16_STAMP(i) do { if (blockIdx.x == 0 && wave == 0 && lane == 0 && it < 64) vw16_dbg[it * 5 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VW16_STAMP(i) do { } while (0)
#endif

struct Vae16Args {
    const float* w16[DRL_MAX_VAE_LAYERS];   // layer 0: [slab][T 8][kg 2][64][4]; last: [slab][T2 2][Tin 8][64][4];
                                            // middle: [ot][Tin][r 4][64]
    const float* b16[DRL_MAX_VAE_LAYERS];   // [ot][q 4][r 4] (layer 0, middle); last layer: natural order, slab-padded
    int mid_floats;                         // floats from w16[1] to the end of b16[n-2] (the middle layers' weights and
                                            // biases, contiguous in the pack), a multiple of 4; 0 = too large for the ring
    const float* wlv16;                     // the logvar head in a middle layer's layout, or null (no eps: the mean path);
    const float* blv16;                     // only vae_wide16s_kernel reads them
};

__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void act4(f32x4& v, int act) {
    if (act == 1) { for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.0f); }
    else if (act == 2) { for (int r = 0; r < 4; ++r) v[r] = fast_sigmoid(v[r]); }
    else if (act == 3) { for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[r])); }
}
// `src` is the same for every thread (a slab of the weight pack): base + one of four per-thread byte offsets that never
// change.  (A VALU instruction of a wave whose SIMD-mate is streaming fp32 MFMAs waits for the MFMA in flight almost
// every time — ~50 cycles each, tools/pp_stamps.py — so a load phase wants as little address arithmetic as possible;
// hipcc still builds the 64-bit vector address with four adds rather than the scalar-base form.)
__device__ __forceinline__ void vw16_stage_load(f32x4 (&reg)[4], const float* __restrict__ src, const unsigned (&voff)[4]) {
#if VW16_ABL == 3 || VW16_ABL == 11
    return;
#endif
    const char* __restrict__ p = reinterpret_cast<const char*>(src);
#pragma unroll
    for (int i = 0; i < 4; ++i) reg[i] = *reinterpret_cast<const f32x4*>(p + voff[i]);
}
__device__ __forceinline__ void vw16_stage_store(const f32x4 (&reg)[4], float* dst) {
#if VW16_ABL == 5 || VW16_ABL == 11
    return;
#endif
    f32x4* p = reinterpret_cast<f32x4*>(dst) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i * 256] = reg[i];
}

// Last-layer epilogue of one 16-row output tile (4 values per lane): activation, input transform, squared error.
template <bool SCALED, int ACT, bool MASKED>
__device__ __forceinline__ float vw16_epilogue(const f32x4& acc, const float (&xv)[4], const f32x4 (&sc)[4], float clip,
                                               int d0, int D, float sq) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float y = apply_act(acc[r], ACT);
        float x = xv[r];
        if (SCALED && VW16_ABL != 1 && VW16_ABL != 11) x = robust_apply(x, sc[0][r], sc[1][r], sc[2][r], sc[3][r], clip);
        float df = y - x;
        if (MASKED) df = (d0 + r < D) ? df : 0.0f;          // the element tail of the last slab is padding
        sq = fmaf(df, df, sq);
    }
    return sq;
}

// (A 168-register build, three workgroups per CU where their LDS fits — SMD-sized regions — spills 25 registers and
// measured 0.186 ms against 0.175 ms on C3: not kept.)
template <bool SCALED>
__global__ __launch_bounds__(256, 2) void vae_wide16_kernel(VaeArgs a, Vae16Args b, int xr_floats) {
    extern __shared__ __attribute__((aligned(16))) float vw16_smem[];
    float* wt = vw16_smem;                          // [2][VW16_SLAB]
    float* side = vw16_smem + 2 * VW16_SLAB;        // [2][VW16_SIDE]
    float* xr = side + 2 * VW16_SIDE;               // [xr_floats]: the workgroup's windows as one flat region
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, col = lane & 15;
    const float clip = a.clip > 0.0f ? a.clip : 3.0e38f;
    const int D = a.D;
    const int tile = blockIdx.x;                    // < a.n_tiles by construction of the grid

    f32x4 stg[4];
    const unsigned voff[4] = {threadIdx.x * 16u, threadIdx.x * 16u + 4096u, threadIdx.x * 16u + 8192u, threadIdx.x * 16u + 12288u};
    vw16_stage_load(stg, b.w16[0], voff);           // layer 0's first slab: in flight while the windows are staged

    // ---- this workgroup's 64 windows: stage the flat region they span, once
    const int s = find_seg(a.seg.tile_off, a.seg.n_seg, tile);
    const long long samp0 = a.seg.samp_off[s];
    const long long nsamp = a.seg.samp_off[s + 1] - samp0;
    const long long Ks = a.seg.win_off[s + 1] - a.seg.win_off[s];
    const long long k0 = (long long)(tile - a.seg.tile_off[s]) * VW16_WIN;
    const int count_wg = (int)(Ks - k0 < VW16_WIN ? Ks - k0 : VW16_WIN);
    {
        const bool has_norm = a.seg.in_scale != nullptr;
        const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
        const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
        // clamped to the series, zero beyond it (the padded element tail of the last slab is read, against
        // zero weights, so it must be finite)
        const long long e0 = samp0 * a.win_stride + a.lag_elems + k0 * a.win_stride;
        const long long e_end = (samp0 + nsamp) * a.win_stride;
        const int len = (int)((count_wg - 1) * a.win_stride + D);
        const float* __restrict__ src = a.x + e0;
        for (int i0 = threadIdx.x; i0 < (VW16_ABL == 9 ? 256 : xr_floats); i0 += 256 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                v[u] = (i < len && e0 + i < e_end) ? src[i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * 256;
                if (i < xr_floats) xr[i] = has_norm ? minmax_apply(v[u], sc, of) : v[u];
            }
        }
    }
    int count = count_wg - 16 * wave;
    count = count < 0 ? 0 : (count > 16 ? 16 : count);
    const bool active = count > 0;                  // wave-uniform; idle waves still stage and sync
    const int jc = col < count ? col : (count > 0 ? count - 1 : 0);
    const float* xw = xr + (size_t)(16 * wave + jc) * a.win_stride;      // this lane's window inside the region
    const long long wout = a.seg.win_off[s] + k0 + 16 * wave + col;

    f32x4 cur[8], nxt[8];

    // ---- layer 0: D -> hidden (8 row tiles of 16), weights streamed per 32-element slab
    {
        const int nin = (VW16_ABL == 6 || VW16_ABL == 8 || VW16_ABL == 9) ? 1 : (a.din[0] + 31) >> 5;
        const float* __restrict__ w = b.w16[0];
#pragma unroll
        for (int T = 0; T < 8; ++T) cur[T] = *reinterpret_cast<const f32x4*>(b.b16[0] + (T * 4 + q) * 4);
        // side tile of a slab, layer 0: the four scaler tables, each slab of 32 columns transposed to [q][ks]
        // (position q*8 + ks holds column 4*ks + q) so that a lane's eight k-steps are two 16-byte reads;
        // every thread below 128 owns one position of one table and streams it slab by slab
        const bool side_thr = SCALED && threadIdx.x < 128;
        const unsigned soff = side_thr ? (unsigned)(((size_t)(threadIdx.x >> 5) * a.rsP + 4 * (threadIdx.x & 7) + ((threadIdx.x & 31) >> 3)) * 4) : 0u;
        const char* __restrict__ rsb = reinterpret_cast<const char*>(SCALED ? a.rs : w);      // the same for every thread
        vw16_stage_store(stg, wt);
        if (side_thr) side[threadIdx.x] = *reinterpret_cast<const float*>(rsb + soff);
        // The biases above came from global memory.  Unless they are consumed HERE, the compiler's wait-count pass
        // keeps "cur[T] may still be in flight" alive around the loop's back edge and puts s_waitcnt vmcnt(7..0) in
        // front of the first eight MFMAs of EVERY slab — which then also wait for the loads of the next slab issued
        // a few instructions earlier (a full L2 round trip exposed per slab instead of hidden under 64 MFMAs).
#pragma unroll
        for (int T = 0; T < 8; ++T) asm volatile("" : "+v"(cur[T]));
        __syncthreads();                            // also: the region is complete
        // Every address of the loop is a value that never changes plus a compile-time offset (the ring buffer index
        // is the unrolled iteration's parity) or a scalar (the slab's base): a slab costs ONE vector instruction
        // besides its loads, MFMAs and the input transform (the walk of xt).
        const float* __restrict__ wl = wt + lane * 4;           // this lane's A fragments inside a ring buffer
        const float* __restrict__ sl = side + q * 8;
        float* const sdw = side + threadIdx.x;
        const float* __restrict__ xt = xw + q;
        auto slab = [&](auto CBt, int it) __attribute__((always_inline)) {
            constexpr int cb = decltype(CBt)::value;
            const bool more = it + 1 < nin;
            float side_next = 0.0f;
            VW16_STAMP(0);
            if (more) {
                if (side_thr) side_next = *reinterpret_cast<const float*>(rsb + (size_t)(it + 1) * 128 + soff);
                vw16_stage_load(stg, w + (size_t)(it + 1) * VW16_SLAB, voff);
            }
            if (active) {
                const float* __restrict__ wb = wl + cb * VW16_SLAB;
                const float* __restrict__ sb = sl + cb * VW16_SIDE;
                // LDS reads in two batches, each requested before the MFMAs of the batch before it: the
                // compiler left to itself reads four fragments and waits for them, every 16 MFMAs
                f32x4 av[16], sc[4][2];
                float xv[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) xv[ks] = xt[4 * ks];
                if (SCALED) {
#pragma unroll
                    for (int tb = 0; tb < 4; ++tb) {
                        sc[tb][0] = *reinterpret_cast<const f32x4*>(sb + tb * 32);
                        sc[tb][1] = *reinterpret_cast<const f32x4*>(sb + tb * 32 + 4);
                    }
                }
#pragma unroll
                for (int T = 0; T < 8; ++T) av[T * 2] = VW16_ABL >= 10 ? f32x4{xv[0], xv[1], xv[2], xv[3]} : *reinterpret_cast<const f32x4*>(wb + (T * 2) * 256);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int T = 0; T < 8; ++T) av[T * 2 + 1] = VW16_ABL >= 10 ? f32x4{xv[4], xv[5], xv[6], xv[7]} : *reinterpret_cast<const f32x4*>(wb + (T * 2 + 1) * 256);
                __builtin_amdgcn_sched_barrier(0);
                VW16_STAMP(1);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    float bv = xv[ks];
                    if (SCALED && VW16_ABL != 1 && VW16_ABL != 11) bv = robust_apply(bv, sc[0][ks >> 2][ks & 3], sc[1][ks >> 2][ks & 3],
                                                  sc[2][ks >> 2][ks & 3], sc[3][ks >> 2][ks & 3], clip);
#pragma unroll
                    for (int T = 0; T < 8; ++T) cur[T] = mfma16x4(av[T * 2 + (ks >> 2)][ks & 3], bv, cur[T]);
                    if (ks == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            xt += 32;
            VW16_STAMP(2);
            if (more) {
                vw16_stage_store(stg, wt + (cb ^ 1) * VW16_SLAB);
                if (side_thr) sdw[(cb ^ 1) * VW16_SIDE] = side_next;
            }
            VW16_STAMP(3);
            VW16_SYNC();
            VW16_STAMP(4);
        };
        int it = 0;
        for (; it + 1 < nin; it += 2) {
            slab(std::integral_constant<int, 0>{}, it);
            slab(std::integral_constant<int, 1>{}, it + 1);
        }
        if (it < nin) slab(std::integral_constant<int, 0>{}, it);
#pragma unroll
        for (int T = 0; T < 8; ++T) act4(cur[T], a.act[0]);
    }

    // ---- middle layers: registers -> registers.  Their weights are small (128 -> 10 -> 128: 16 KiB) but a wave
    // that fetches each fragment from L2 as it needs it pays a round trip per MFMA (tens of microseconds
    // per workgroup).  The ring is idle between the two streamed layers: the workgroup copies all of
    // them into it with every load in flight, and the fragments come from LDS.  The last layer's first slab
    // is requested before, and lands in registers meanwhile.
    const int last = a.n - 1;
    vw16_stage_load(stg, b.w16[last], voff);
    const bool mid_lds = b.mid_floats > 0;
    if (mid_lds) {
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(b.w16[1]);
        f32x4* dst = reinterpret_cast<f32x4*>(wt);
        const int n4 = b.mid_floats >> 2;
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (threadIdx.x + u * 256 < n4) t[u] = src[threadIdx.x + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (threadIdx.x + u * 256 < n4) dst[threadIdx.x + u * 256] = t[u];
        __syncthreads();
    }
    for (int p = 1; p < last; ++p) {
        const int nin = (a.din[p] + 15) >> 4, nout = (a.dout[p] + 15) >> 4;
        const float* __restrict__ w = mid_lds ? wt + (b.w16[p] - b.w16[1]) : b.w16[p];
        const float* __restrict__ bb = mid_lds ? wt + (b.b16[p] - b.w16[1]) : b.b16[p];
        if (active) {
#pragma unroll
            for (int ot = 0; ot < 8; ++ot)
                nxt[ot] = ot < nout ? *reinterpret_cast<const f32x4*>(bb + (ot * 4 + q) * 4) : f32x4{};
#pragma unroll
            for (int Tin = 0; Tin < 8; ++Tin) {
                if (Tin < nin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int ot = 0; ot < 8; ++ot)
                            if (ot < nout)
                                nxt[ot] = mfma16x4(w[(((size_t)ot * nin + Tin) * 4 + r) * 64 + lane], cur[Tin][r], nxt[ot]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // one input tile's fragments in flight at a time
            }
#pragma unroll
            for (int ot = 0; ot < 8; ++ot) { cur[ot] = nxt[ot]; act4(cur[ot], a.act[p]); }
        }
    }
    if (mid_lds) __syncthreads();                   // the ring is rewritten below

    // ---- last layer: hidden -> D, weights streamed per 32-element output slab, error fused
    {
        const int p = a.n - 1;
        const int nout = (VW16_ABL == 7 || VW16_ABL == 8 || VW16_ABL == 9) ? 1 : (a.dout[p] + 31) >> 5;
        const float* __restrict__ w = b.w16[p];
        const int act = a.act[p];
        float sq = 0.0f;
        // side tile, last layer: the scaler tables in natural column order, then the slab's 32 biases
        const float* __restrict__ ss = (SCALED && threadIdx.x < 128)
            ? a.rs + (size_t)(threadIdx.x >> 5) * a.rsP + (threadIdx.x & 31)
            : (threadIdx.x >= 128 && threadIdx.x < 160) ? b.b16[p] + (threadIdx.x - 128) : w;
        vw16_stage_store(stg, wt);                  // requested before the middle layers
        if (threadIdx.x < VW16_SIDE) side[threadIdx.x] = ss[0];
        __syncthreads();
        for (int ot = 0; ot < nout; ++ot) {
            const int cb = ot & 1;
            const bool more = ot + 1 < nout;
            float side_next = 0.0f;
            if (more) {
                side_next = ss[(ot + 1) * 32];
                vw16_stage_load(stg, w + (size_t)(ot + 1) * VW16_SLAB, voff);
            }
            if (active) {
                const float* __restrict__ wb = wt + cb * VW16_SLAB + lane * 4;
                const float* __restrict__ sb = side + cb * VW16_SIDE + 4 * q;
                const float* __restrict__ xt = xw + ot * 32 + 4 * q;
                f32x4 acc0 = *reinterpret_cast<const f32x4*>(sb + 128);
                f32x4 acc1 = *reinterpret_cast<const f32x4*>(sb + 128 + 16);
                f32x4 av[16], sc[2][4];
                float xv[2][4];
                // LDS reads in two batches, the second (with the epilogue's operands) under the MFMAs of the first
#pragma unroll
                for (int Tin = 0; Tin < 4; ++Tin) {
                    av[Tin] = VW16_ABL >= 10 ? cur[Tin] : *reinterpret_cast<const f32x4*>(wb + Tin * 256);
                    av[8 + Tin] = VW16_ABL >= 10 ? cur[7 - Tin] : *reinterpret_cast<const f32x4*>(wb + (8 + Tin) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int Tin = 4; Tin < 8; ++Tin) {
                    av[Tin] = VW16_ABL >= 10 ? cur[Tin] : *reinterpret_cast<const f32x4*>(wb + Tin * 256);
                    av[8 + Tin] = VW16_ABL >= 10 ? cur[7 - Tin] : *reinterpret_cast<const f32x4*>(wb + (8 + Tin) * 256);
                }
#pragma unroll
                for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xv[T2][r] = xt[16 * T2 + r];
                if (SCALED) {
#pragma unroll
                    for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
                        for (int tb = 0; tb < 4; ++tb) sc[T2][tb] = *reinterpret_cast<const f32x4*>(sb + tb * 32 + 16 * T2);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int Tin = 0; Tin < 8; ++Tin) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc0 = mfma16x4(av[Tin][r], cur[Tin][r], acc0);
                        acc1 = mfma16x4(av[8 + Tin][r], cur[Tin][r], acc1);
                    }
                    if (Tin == 3) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // activation code and "is this the padded last slab" are wave-uniform: one straight-line
                // epilogue per combination
                const bool masked = ot + 1 == nout;
                const int d0 = ot * 32 + 4 * q;
#define VW16_EPI(ACT)                                                                                        \
                if (masked) { sq = vw16_epilogue<SCALED, ACT, true>(acc0, xv[0], sc[0], clip, d0, D, sq);        \
                              sq = vw16_epilogue<SCALED, ACT, true>(acc1, xv[1], sc[1], clip, d0 + 16, D, sq); } \
                else { sq = vw16_epilogue<SCALED, ACT, false>(acc0, xv[0], sc[0], clip, d0, D, sq);             \
                       sq = vw16_epilogue<SCALED, ACT, false>(acc1, xv[1], sc[1], clip, d0 + 16, D, sq); }
#if VW16_ABL == 2 || VW16_ABL == 11
                sq += acc0[0] + acc1[3] + xv[0][0] + xv[1][3] + (SCALED ? sc[0][0][0] + sc[1][3][3] : 0.0f) + (masked ? (float)d0 : 0.0f);
#else
                if (act == 3) { VW16_EPI(3) } else if (act == 2) { VW16_EPI(2) } else if (act == 1) { VW16_EPI(1) } else { VW16_EPI(0) }
#endif
#undef VW16_EPI
            }
            if (more) {
                vw16_stage_store(stg, wt + (cb ^ 1) * VW16_SLAB);
                if (threadIdx.x < VW16_SIDE) side[(cb ^ 1) * VW16_SIDE + threadIdx.x] = side_next;
            }
            VW16_SYNC();
        }
        sq += __shfl_xor(sq, 16);                   // the four q of a window hold different rows
        sq += __shfl_xor(sq, 32);
        if (active && a.err && q == 0 && col < count) a.err[wout] = sq / (float)D;
    }
}
