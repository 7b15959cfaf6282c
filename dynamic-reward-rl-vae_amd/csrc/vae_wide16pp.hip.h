// K1+K2 for wide windows, large calls: vae_wide16_kernel's arithmetic (same packed weights, same MFMA order per
// accumulator, same bits) as a PING-PONG of two 64-window tiles on one CU.
//
// Why.  fp32 MFMA runs at the vector rate and shares the SIMD's vector issue with every other VALU instruction, and a
// wave cannot have LDS reads in flight for the slab after the one it is multiplying without twice the registers.  In
// vae_wide16_kernel the two waves of a SIMD belong to two independent workgroups, each with its own copy of the
// weight ring; nothing keeps their phases apart, and whenever both sit in their load phase (barrier, 28 LDS reads and
// their latency, staging stores) the matrix pipe idles: 65 % MFMA-busy on WaDi shapes.  Here one workgroup of 8 waves
// holds TWO tiles (group A = waves 0-3, group B = waves 4-7; one wave of each per SIMD) over ONE weight ring, and the
// workgroup barrier alternates their roles: while A multiplies slab i out of registers — a phase of nothing but 64
// MFMAs — B reads its fragments of slab i from LDS, transforms its inputs, finishes the epilogue of the slab before
// and stages the ring; then they swap.  A SIMD's matrix pipe always has exactly one wave feeding it.  The weights are
// streamed once per 128 windows instead of once per 64 (half the L2 -> LDS traffic), the grid is persistent with a
// balanced share of tiles per CU (an odd tile at the end runs alone, in half the time), and the window regions are
// staged with 16-byte loads.
//
// Layouts, packing and the per-window arithmetic: vae_wide16.hip.h.  Results are bit-identical to that kernel (tested).
#pragma once

#define VWP_THREADS 512
// VWP_STAMPS (diagnostic builds only): s_memtime stamps of the first half-steps of layer 0, workgroup 0, waves 0 and 4
#ifdef VWP_STAMPS
__device__ unsigned long long vwp_dbg[2 * 64 * 3];
#define VWP_STAMP(i) do { if (blockIdx.x == 0 && pair == t_lo && w4 == 0 && lane == 0 && h < 48) \
    vwp_dbg[(grp * 64 + h) * 3 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define VWP_MARK(slot) do { if (blockIdx.x == 0 && pair == t_lo && w4 == 0 && lane == 0) \
    vwp_dbg[(grp * 64 + 60) * 3 - 24 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define VWP_STAMP(i) do { } while (0)
#define VWP_MARK(slot) do { } while (0)
#endif
#ifndef VWP_ABL
#define VWP_ABL 0            // timing-only ablations of layer 0's load phase (wrong results): 1 no transform, 2 no A-fragment reads
#endif
#ifndef VWP_PRIO
#define VWP_PRIO 3           // wave priority of a load phase (the MFMA phase of the SIMD's other wave runs at 0)
#endif

// One 16 KiB weight slab, global -> LDS by LDS-DMA (global_load_lds_dwordx4: no register destination, no ds_write, and
// the compiler drains it with vmcnt(0) at the workgroup barrier that ends the phase).  Issued by the 256 threads of the
// staging group: instruction i of wave w4 writes the 1 KiB at float offset i*1024 + w4*256 of the ring buffer (LDS
// destination = wave-uniform base + lane*16), from per-lane source pointers that are advanced in the group's MFMA
// phases, so that a load phase spends no vector instruction on addresses.
typedef __attribute__((address_space(1))) const void* vwp_gptr;
typedef __attribute__((address_space(3))) void* vwp_lptr;
__device__ __forceinline__ void vwp_dma_slab(const float* const (&src)[4], float* dst_buf, int w4) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((vwp_gptr)src[i], (vwp_lptr)(dst_buf + i * 1024 + w4 * 256), 16, 0, 0);
}
// the slab's side tile: one dword per thread of the staging group's first 128 (160) threads
__device__ __forceinline__ void vwp_dma_side(const float* src, float* dst_side, int w4) {
    __builtin_amdgcn_global_load_lds((vwp_gptr)src, (vwp_lptr)(dst_side + w4 * 64), 4, 0, 0);
}

// The MFMA with its accumulator tied (dst = srcC), as an asm statement: the builtin's untied form makes the register
// allocator accumulate out of place and copy all 32 accumulator registers back after every phase.  `nops` wait states
// in front cover the one hazard the compiler cannot see inside an asm statement: a VALU result (the transformed
// input) used as an MFMA operand straight away.  Accumulators are only read by VALU code one barrier later.
__device__ __forceinline__ void vwp_mfma_inplace(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void vwp_mfma_inplace_nop(f32x4& acc, float a, float b) {
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// A VALU instruction of a wave whose SIMD-mate is streaming fp32 MFMAs waits for the MFMA in flight (32 cycles) almost
// every time: measured ~50 cycles each (tools/pp_stamps.py: a load phase with the 40 instructions of the input
// transform took 2,600 cycles, without them 530).  So a load phase is LDS reads and nothing else; every VALU
// instruction a slab needs — input transform, epilogue — is issued by the wave that multiplies it, between its own
// MFMAs, at 4 cycles each.

// layer 0, load phase of one slab: A fragments, this lane's eight raw inputs and their scaler entries, LDS -> registers
template <bool SCALED>
__device__ __forceinline__ void vwp_l0_load(f32x4 (&av)[16], float (&xv)[8], f32x4 (&sc)[4][2], const float* __restrict__ wb,
                                            const float* __restrict__ sb, const float* __restrict__ xt) {
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
    for (int i = 0; i < 16; ++i) av[i] = *reinterpret_cast<const f32x4*>(wb + i * 256);
}
// layer 0, MFMA phase: the transform of input ks, then the eight MFMAs it feeds
template <bool SCALED>
__device__ __forceinline__ void vwp_l0_mfma(f32x4 (&cur)[8], const f32x4 (&av)[16], const float (&xv)[8], const f32x4 (&sc)[4][2],
                                            float clip) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        float bv = xv[ks];
        if (SCALED && !(VWP_ABL & 1))
            bv = robust_apply(bv, sc[0][ks >> 2][ks & 3], sc[1][ks >> 2][ks & 3], sc[2][ks >> 2][ks & 3], sc[3][ks >> 2][ks & 3], clip);
        vwp_mfma_inplace_nop(cur[0], av[ks >> 2][ks & 3], bv);
#pragma unroll
        for (int T = 1; T < 8; ++T) vwp_mfma_inplace(cur[T], av[T * 2 + (ks >> 2)][ks & 3], bv);
    }
}

// last layer, load phase of one output slab: fragments, the two accumulators' biases, the eight raw inputs the slab's
// outputs are compared with and their scaler entries
template <bool SCALED>
__device__ __forceinline__ void vwp_ll_load(f32x4 (&av)[16], f32x4& accn0, f32x4& accn1, float (&xv)[8], f32x4 (&sc)[2][4],
                                            const float* __restrict__ wb, const float* __restrict__ sb,
                                            const float* __restrict__ xt) {
    accn0 = *reinterpret_cast<const f32x4*>(sb + 128);
    accn1 = *reinterpret_cast<const f32x4*>(sb + 128 + 16);
#pragma unroll
    for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
        for (int r = 0; r < 4; ++r) xv[T2 * 4 + r] = xt[16 * T2 + r];
    if (SCALED) {
#pragma unroll
        for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) sc[T2][tb] = *reinterpret_cast<const f32x4*>(sb + tb * 32 + 16 * T2);
    }
#pragma unroll
    for (int Tin = 0; Tin < 8; ++Tin) {
        av[Tin] = *reinterpret_cast<const f32x4*>(wb + Tin * 256);
        av[8 + Tin] = *reinterpret_cast<const f32x4*>(wb + (8 + Tin) * 256);
    }
}
template <bool SCALED>
__device__ __forceinline__ void vwp_ll_transform(float (&xt8)[8], const float (&xv)[8], const f32x4 (&sc)[2][4], float clip) {
#pragma unroll
    for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            xt8[T2 * 4 + r] = SCALED ? robust_apply(xv[T2 * 4 + r], sc[T2][0][r], sc[T2][1][r], sc[T2][2][r], sc[T2][3][r], clip)
                                     : xv[T2 * 4 + r];
}
__device__ __forceinline__ void vwp_ll_mfma(f32x4& acc0, f32x4& acc1, const f32x4 (&av)[16], const f32x4 (&cur)[8]) {
#pragma unroll
    for (int Tin = 0; Tin < 8; ++Tin)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (Tin == 0 && r == 0) vwp_mfma_inplace_nop(acc0, av[0][0], cur[0][0]);      // acc0 / acc1 were just copied by VALU moves
            else vwp_mfma_inplace(acc0, av[Tin][r], cur[Tin][r]);
            vwp_mfma_inplace(acc1, av[8 + Tin][r], cur[Tin][r]);
        }
}
// epilogue of one output slab: activation, difference to the (transformed) input, running sum of squares — the
// operations and the order of vw16_epilogue
template <int ACT>
__device__ __forceinline__ float vwp_ll_epi_ct(const f32x4& acc0, const f32x4& acc1, const float (&xt8)[8], bool masked,
                                               int d0, int D, float sq) {
#pragma unroll
    for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float y = apply_act(T2 ? acc1[r] : acc0[r], ACT);
            float df = y - xt8[T2 * 4 + r];
            if (masked) df = (d0 + 16 * T2 + r < D) ? df : 0.0f;
            sq = fmaf(df, df, sq);
        }
    return sq;
}
__device__ __forceinline__ float vwp_ll_epi(const f32x4& acc0, const f32x4& acc1, const float (&xt8)[8], int act, bool masked,
                                            int d0, int D, float sq) {
    if (act == 3) return vwp_ll_epi_ct<3>(acc0, acc1, xt8, masked, d0, D, sq);
    if (act == 2) return vwp_ll_epi_ct<2>(acc0, acc1, xt8, masked, d0, D, sq);
    if (act == 1) return vwp_ll_epi_ct<1>(acc0, acc1, xt8, masked, d0, D, sq);
    return vwp_ll_epi_ct<0>(acc0, acc1, xt8, masked, d0, D, sq);
}

template <bool SCALED>
__global__ __launch_bounds__(VWP_THREADS, 1) void vae_wide16pp_kernel(VaeArgs a, Vae16Args b, int xr_floats) {
    extern __shared__ __attribute__((aligned(16))) float vwp_smem[];
    float* wt = vwp_smem;                               // [2][VW16_SLAB]   the weight ring, shared by both groups
    float* side = vwp_smem + 2 * VW16_SLAB;             // [2][VW16_SIDE]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;     // scalar: uniform branches
    const int grp = wave >> 2, w4 = wave & 3, gt = threadIdx.x & 255;     // group, wave and thread inside the group
    float* xr = side + 2 * VW16_SIDE + (size_t)grp * xr_floats;          // the group's windows as one flat region
    const int q = lane >> 4, col = lane & 15;
    const float clip = a.clip > 0.0f ? a.clip : 3.0e38f;
    const int D = a.D;
    const bool stager = grp == 1;                       // group B streams the ring (in its load phases)
    // this workgroup's share of the tiles: balanced, contiguous
    const int t_lo = (int)((long long)blockIdx.x * a.n_tiles / gridDim.x);
    const int t_hi = (int)((long long)(blockIdx.x + 1) * a.n_tiles / gridDim.x);
    const int last = a.n - 1;

    for (int pair = t_lo; pair < t_hi; pair += 2) {
        const int tile = pair + grp;
        const bool gact = tile < t_hi;                  // group-uniform: an odd tile at the end runs alone
        VWP_MARK(0);
        if (stager) {                                   // layer 0's first slab: lands while the windows are staged
            const float* const p0[4] = {b.w16[0] + gt * 4, b.w16[0] + gt * 4 + 1024, b.w16[0] + gt * 4 + 2048, b.w16[0] + gt * 4 + 3072};
            vwp_dma_slab(p0, wt, w4);
        }

        // ---- the group's 64 windows: the flat region they span, staged once, 16 bytes per load
        int count_wg = 0, s = 0, sh = 0;
        long long k0 = 0;
        if (gact) {
            s = find_seg(a.seg.tile_off, a.seg.n_seg, tile);
            const long long samp0 = a.seg.samp_off[s];
            const long long nsamp = a.seg.samp_off[s + 1] - samp0;
            const long long Ks = a.seg.win_off[s + 1] - a.seg.win_off[s];
            k0 = (long long)(tile - a.seg.tile_off[s]) * VW16_WIN;
            count_wg = (int)(Ks - k0 < VW16_WIN ? Ks - k0 : VW16_WIN);
            const bool has_norm = a.seg.in_scale != nullptr;
            const float nsc = has_norm ? a.seg.in_scale[s] : 1.0f;
            const float nof = has_norm ? a.seg.in_offset[s] : 0.0f;
            const long long e0 = samp0 * a.win_stride + a.lag_elems + k0 * a.win_stride;
            const long long e_end = (samp0 + nsamp) * a.win_stride;
            const int len = (int)((count_wg - 1) * a.win_stride + D);
            sh = (int)(e0 & 3);                         // the series base is 16-byte aligned (host): region[i] = element e0-sh+i
            const float* __restrict__ src = a.x + (e0 - sh);
            const long long room = e_end - (e0 - sh);   // elements of the series from the region's first one on
            for (int i0 = gt * 4; i0 < xr_floats; i0 += 256 * 4 * 4) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 1024;
                    // clamped to the series, zero beyond it (the padded element tail of the last slab is read,
                    // against zero weights, so it must be finite)
                    if (i + 3 - sh < len && i + 3 < room) v[u] = *reinterpret_cast<const f32x4*>(src + i);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] = (i + e - sh < len && i + e < room) ? src[i + e] : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 1024;
                    if (i < xr_floats) {
                        if (has_norm) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[u][e] = minmax_apply(v[u][e], nsc, nof);
                        }
                        *reinterpret_cast<f32x4*>(xr + i) = v[u];
                    }
                }
            }
        }
        int count = count_wg - 16 * w4;
        count = count < 0 ? 0 : (count > 16 ? 16 : count);
        const bool active = count > 0;                  // wave-uniform; idle waves still stage and sync
        const int jc = col < count ? col : (count > 0 ? count - 1 : 0);
        const float* xw = xr + sh + (size_t)(16 * w4 + jc) * a.win_stride;      // this lane's window inside the region
        const long long wout = a.seg.win_off[s] + k0 + 16 * w4 + col;

        f32x4 cur[8], av[16];

        // ---- layer 0: D -> hidden, ping-pong over the 32-element slabs
        {
            const int nin = (a.din[0] + 31) >> 5;
            const float* __restrict__ w = b.w16[0];
#pragma unroll
            for (int T = 0; T < 8; ++T) cur[T] = *reinterpret_cast<const f32x4*>(b.b16[0] + (T * 4 + q) * 4);
            // side tile of a slab: the four scaler tables, 32 columns transposed to [q][ks] (vae_wide16.hip.h)
            const bool side_thr = SCALED && stager && gt < 128;
            const float* sp = side_thr ? a.rs + (size_t)(gt >> 5) * a.rsP + 4 * (gt & 7) + ((gt & 31) >> 3) : w;
            // source pointers of the NEXT slab to stage (slab 1 now), advanced as each DMA is issued
            const float* dp[4] = {w + VW16_SLAB + gt * 4, w + VW16_SLAB + gt * 4 + 1024, w + VW16_SLAB + gt * 4 + 2048,
                                  w + VW16_SLAB + gt * 4 + 3072};
            if (side_thr) vwp_dma_side(sp, side, w4);
            sp += 32;
            // the biases are consumed here, not by the first MFMA inside the loop (see vae_wide16.hip.h)
#pragma unroll
            for (int T = 0; T < 8; ++T) asm volatile("" : "+v"(cur[T]));
            __syncthreads();                            // slab 0 and both regions are in LDS
            float xv[8];
            f32x4 sc[4][2];
            // Half-steps, one barrier each.  Group g loads slab it at half-step 2*it + g and multiplies it at half-step
            // 2*it + g + 1: while A multiplies slab it (half-step 2*it + 1), B starts the DMA of slab it+1 into the ring and
            // loads its own operands of slab it; while B multiplies it, A loads its operands of slab it+1.  Waves without
            // windows run along on whatever their lanes' clamped addresses hold (no result of theirs is stored): every
            // value below is then written in exactly one branch, unconditionally, which keeps the compiler from copying
            // the accumulators around.
            VWP_MARK(1);
            if (grp == 0) vwp_l0_load<SCALED>(av, xv, sc, wt + lane * 4, side + q * 8, xw + q);
            __syncthreads();
            for (int h = 1; h <= 2 * nin; ++h) {
                const int k = h - grp;
                VWP_STAMP(0);
                if (k & 1) {
                    vwp_l0_mfma<SCALED>(cur, av, xv, sc, clip);
                } else {
                    const int it = k >> 1, cb = it & 1;
                    // (a no-op "definition" of the accumulators on this path too: with a new value on both paths the
                    // register allocator keeps the MFMA phase's accumulation in place instead of copying 32 registers)
#pragma unroll
                    for (int T = 0; T < 8; ++T) asm volatile("" : "+v"(cur[T]));
                    if (it < nin) {
                        vwp_l0_load<SCALED>(av, xv, sc, wt + cb * VW16_SLAB + lane * 4, side + cb * VW16_SIDE + q * 8,
                                            xw + it * 32 + q);
                        // the DMA after the group's own reads: the compiler orders an LDS read behind a DMA in flight
                        if (stager && it + 1 < nin) {           // slab it+1 into the buffer slab it-1 has left
                            vwp_dma_slab(dp, wt + (cb ^ 1) * VW16_SLAB, w4);
                            if (side_thr) vwp_dma_side(sp, side + (cb ^ 1) * VW16_SIDE, w4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) dp[i] += VW16_SLAB;
                            sp += 32;
                        }
                    }
                }
                VWP_STAMP(1);
                __syncthreads();
                VWP_STAMP(2);
            }
#pragma unroll
            for (int T = 0; T < 8; ++T) act4(cur[T], a.act[0]);
        }

        VWP_MARK(2);
        // ---- middle layers: registers -> registers, both groups at once; their weights through the idle ring
        const bool mid_lds = b.mid_floats > 0;
        if (mid_lds) {
            const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(b.w16[1]);
            f32x4* dst = reinterpret_cast<f32x4*>(wt);
            const int n4 = b.mid_floats >> 2;
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) if ((int)threadIdx.x + u * VWP_THREADS < n4) t[u] = src[threadIdx.x + u * VWP_THREADS];
#pragma unroll
            for (int u = 0; u < 4; ++u) if ((int)threadIdx.x + u * VWP_THREADS < n4) dst[threadIdx.x + u * VWP_THREADS] = t[u];
            __syncthreads();
        }
        for (int p = 1; p < last; ++p) {
            const int nin = (a.din[p] + 15) >> 4, nout = (a.dout[p] + 15) >> 4;
            const float* __restrict__ w = mid_lds ? wt + (b.w16[p] - b.w16[1]) : b.w16[p];
            const float* __restrict__ bb = mid_lds ? wt + (b.b16[p] - b.w16[1]) : b.b16[p];
            {
                f32x4 nxt[8];
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
                }
#pragma unroll
                for (int ot = 0; ot < 8; ++ot) { cur[ot] = nxt[ot]; act4(cur[ot], a.act[p]); }
            }
        }
        if (mid_lds) __syncthreads();                   // the ring is rewritten below

        // ---- last layer: hidden -> D per 32-element output slab, error fused; the epilogue of a slab runs in the
        // load phase that follows its MFMA phase
        float sq = 0.0f;
        {
            const int nout = (a.dout[last] + 31) >> 5;
            const float* __restrict__ w = b.w16[last];
            const int act = a.act[last];
            // side tile: the scaler tables in natural column order, then the slab's 32 biases
            const float* sp = (SCALED && stager && gt < 128) ? a.rs + (size_t)(gt >> 5) * a.rsP + (gt & 31)
                              : (stager && gt >= 128 && gt < 160) ? b.b16[last] + (gt - 128) : w;
            const bool side_thr = stager && ((SCALED && gt < 128) || (gt >= 128 && gt < 160));
            const float* dp[4] = {w + gt * 4, w + gt * 4 + 1024, w + gt * 4 + 2048, w + gt * 4 + 3072};
            if (stager) {
                vwp_dma_slab(dp, wt, w4);
                if (side_thr) vwp_dma_side(sp, side, w4);
#pragma unroll
                for (int i = 0; i < 4; ++i) dp[i] += VW16_SLAB;
                sp += 32;
            }
            __syncthreads();
            f32x4 acc0 = {}, acc1 = {}, accn0 = {}, accn1 = {};
            float xt8[8] = {}, xv[8] = {};
            f32x4 sc[2][4];
            // the same half-steps.  A load phase brings slab ot's operands into registers (LDS reads only); the MFMA phase
            // that follows first finishes slab ot-1 (activation, squared error), transforms slab ot's inputs, then multiplies
            VWP_MARK(3);
            if (grp == 0)
                vwp_ll_load<SCALED>(av, accn0, accn1, xv, sc, wt + lane * 4, side + 4 * q, xw + 4 * q);
            __syncthreads();
            for (int h = 1; h <= 2 * nout; ++h) {
                const int k = h - grp;
                if (k & 1) {
                    const int ot = (k - 1) >> 1;
                    if (ot > 0) sq = vwp_ll_epi(acc0, acc1, xt8, act, false, (ot - 1) * 32 + 4 * q, D, sq);
                    vwp_ll_transform<SCALED>(xt8, xv, sc, clip);
                    acc0 = accn0; acc1 = accn1;
                    vwp_ll_mfma(acc0, acc1, av, cur);
                } else {
                    const int ot = k >> 1, cb = ot & 1;
                    asm volatile("" : "+v"(acc0), "+v"(acc1));
                    if (ot < nout) {
                        vwp_ll_load<SCALED>(av, accn0, accn1, xv, sc, wt + cb * VW16_SLAB + lane * 4,
                                            side + cb * VW16_SIDE + 4 * q, xw + ot * 32 + 4 * q);
                        if (stager && ot + 1 < nout) {
                            vwp_dma_slab(dp, wt + (cb ^ 1) * VW16_SLAB, w4);
                            if (side_thr) vwp_dma_side(sp, side + (cb ^ 1) * VW16_SIDE, w4);
#pragma unroll
                            for (int i = 0; i < 4; ++i) dp[i] += VW16_SLAB;
                            sp += 32;
                        }
                    }
                }
                __syncthreads();
            }
            if (active) sq = vwp_ll_epi(acc0, acc1, xt8, act, true, (nout - 1) * 32 + 4 * q, D, sq);
        }
        VWP_MARK(4);
        sq += __shfl_xor(sq, 16);                       // the four q of a window hold different rows
        sq += __shfl_xor(sq, 32);
        if (active && a.err && q == 0 && col < count) a.err[wout] = sq / (float)D;
        // the next pair restages the regions and the ring: every wave is past its last LDS read (the loop's last barrier)
    }
}
