// K1+K2 for wide windows, software-pipelined by hand: vae_wide16_kernel's arithmetic (same packed weights, same MFMA
// order per accumulator, same bits) with every memory operation and every input-transform instruction of a slab issued
// from INSIDE the MFMA stream of the slab before it — one of them behind each MFMA.
//
// Why (round 3 measurements: tools/vw16_stamps.py, tools/pp_stamps.py, tools/vws_stamps.py).  In
// vae_wide16_kernel a wave spends, per 32-element slab, ~900 ticks requesting the next slab and reading its 28 LDS
// operands, 2,048+ multiplying, ~600 storing the staged slab and ~100 in the barrier; the second workgroup of the CU is
// meant to multiply meanwhile, but every VALU instruction a wave issues while its SIMD-mate streams fp32 MFMAs waits for
// the MFMA in flight (~50 ticks each), so the non-MFMA phases are long: 65 % MFMA-busy.  (A workgroup barrier that
// alternates two tiles' phases — vae_wide16pp_kernel — keeps the phases apart and not the VALU work: measured no
// faster.)  A wave's OWN loads, LDS reads and stores issue in the shadow of its own MFMAs, but only BETWEEN them: the
// same instructions in a block ahead of the MFMAs cost ~17 ticks each.  So
//   * the slab loop is 64 slots of (MFMA, one other instruction, sched_barrier).  During the first half of slab i the
//     second-half A fragments of slab i and the inputs and scaler entries of slab i+1 are read from LDS; during the
//     second half the first-half fragments of slab i+1, the LDS stores of slab i+2, the global loads of slab i+3 and the
//     input transform of slab i+1 — every value lands in a register the MFMAs have just released (no second register
//     set, no copies);
//   * the body has no branches: past the end of a layer it reads operands nobody uses and writes the ring buffer nobody
//     reads (with branches the compiler's waits count the longest path and the next slab's operands go through 80 v_mov);
//   * a ring of 2 slabs and TWO barriers per slab (the second one between the halves frees slab i's buffer for slab
//     i+2), so that two workgroups still share a CU (77 KB each for WaDi): C4 2.41 ms.  A ring of 3 needs one barrier per
//     slab but fits one workgroup per CU, and then nothing covers region staging, middle layers and barriers: 2.79 ms.
//     Where the region leaves room for one workgroup anyway (more than ~131 sensors at W = 25) the two rings measure
//     the same (F = 140, 131,072 windows: 2.30 / 2.31 ms, vae_wide16_kernel 2.78), so only the ring of 2 is instantiated;
//   * with eps (the reference's sampled z) the middle layers run the logvar head beside mu.
// The host uses this kernel wherever one of its workgroups fits a CU's LDS (knob vae_s = 0: vae_wide16_kernel).
//
// Layouts, packing and the per-window arithmetic: vae_wide16.hip.h.  Results are bit-identical to that kernel (tested).
#pragma once


#ifdef VWS_STAMPS
__device__ unsigned long long vws_dbg[64 * 4];
#define VWS_STAMP(i) do { if (blockIdx.x == 0 && wave == 0 && lane == 0 && it < 60) vws_dbg[it * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define VWS_MARK(i) do { if (blockIdx.x == 0 && wave == 0 && lane == 0) { vws_dbg[240 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0 || (i) == 6) vws_dbg[248 + ((i) != 0)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define VWS_MARK(i) do { } while (0)
#define VWS_STAMP(i) do { } while (0)
#endif

// VWS_RING: slabs in the weight ring.  2: two barriers per slab (shipped); 3: one barrier per slab (A/B builds).
// NT: 16-row tiles of the first and last hidden layer that can be non-zero (8: widths up to 128; 4: up to 64).  The packed
// layout is that of 8 tiles either way (vae_wide16.hip.h, zero-padded); the MFMAs and fragment reads of the tiles beyond
// NT are simply not issued — their products are exact zeros.
template <bool SCALED, int VWS_RING, int NT>
__global__ __launch_bounds__(256, 2) void vae_wide16s_kernel(VaeArgs a, Vae16Args b, int xr_floats) {
    static_assert(VWS_RING == 2 || VWS_RING == 3, "ring of two or three slabs");
    static_assert(NT == 4 || NT == 8, "four or eight row tiles");
    constexpr int SPM = 8 / NT;                             // side operations per MFMA slot of the slab loops
    constexpr int HT = NT / 2;                              // last layer: input tiles per half
    extern __shared__ __attribute__((aligned(16))) float vws_smem[];
    float* wt = vws_smem;                                   // [VWS_RING][VW16_SLAB]
    float* side = vws_smem + VWS_RING * VW16_SLAB;          // [VWS_RING][VW16_SIDE]
    float* xr = side + VWS_RING * VW16_SIDE;                // [xr_floats]: the workgroup's windows as one flat region
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, col = lane & 15;
    const float clip = a.clip > 0.0f ? a.clip : 3.0e38f;
    const int D = a.D;
    const int tile = blockIdx.x;                            // < a.n_tiles by construction of the grid
    const int last = a.n - 1;

    const unsigned voff[4] = {threadIdx.x * 16u, threadIdx.x * 16u + 4096u, threadIdx.x * 16u + 8192u, threadIdx.x * 16u + 12288u};
    f32x4 stg[4], stg2[4];
    vw16_stage_load(stg, b.w16[0], voff);                   // layer 0's first two slabs: in flight while the windows are staged
    if (a.din[0] > 32) vw16_stage_load(stg2, b.w16[0] + VW16_SLAB, voff);

    VWS_MARK(0);
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
        for (int i0 = threadIdx.x; i0 < xr_floats; i0 += 256 * 8) {
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
    VWS_MARK(1);
    int count = count_wg - 16 * wave;
    count = count < 0 ? 0 : (count > 16 ? 16 : count);
    const int jc = col < count ? col : (count > 0 ? count - 1 : 0);
    // (waves without windows run along on their lanes' clamped addresses: nothing of theirs is stored)
    const float* xw = xr + (size_t)(16 * wave + jc) * a.win_stride;      // this lane's window inside the region
    const long long wout = a.seg.win_off[s] + k0 + 16 * wave + col;

    f32x4 cur[8], av[16];
    float* const stw = wt + threadIdx.x * 4;                // where this thread stores its part of a staged slab
    float* const sdw = side + threadIdx.x;
    const float* const wl = wt + lane * 4;                  // this lane's A fragments inside a ring buffer

    // ---- layer 0: D -> hidden (8 row tiles of 16), weights streamed per 32-element slab
    {
        const int nin = (a.din[0] + 31) >> 5;
        const float* __restrict__ w = b.w16[0];
#pragma unroll
        for (int T = 0; T < 8; ++T) cur[T] = *reinterpret_cast<const f32x4*>(b.b16[0] + (T * 4 + q) * 4);
        // side tile of a slab, layer 0: the four scaler tables, each slab of 32 columns transposed to [q][ks]
        // (vae_wide16.hip.h); every thread below 128 owns one position of one table and streams it slab by slab
        const bool side_thr = SCALED && threadIdx.x < 128;
        const float* __restrict__ ss = side_thr ? a.rs + (size_t)(threadIdx.x >> 5) * a.rsP + 4 * (threadIdx.x & 7) + ((threadIdx.x & 31) >> 3) : w;
        float side_next = 0.0f;
        // slabs 0 and 1 into the ring, slab 2 requested
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(stw + i * 1024) = stg[i];
        if (side_thr) sdw[0] = ss[0];
        if (nin > 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(stw + VW16_SLAB + i * 1024) = stg2[i];
            if (side_thr) sdw[VW16_SIDE] = ss[32];
        }
        {
            const int s2 = nin > 2 ? 2 : nin - 1;           // (not beyond the layer: the loop stores what this brings)
            vw16_stage_load(stg, w + (size_t)s2 * VW16_SLAB, voff);
            if (side_thr) side_next = ss[s2 * 32];
        }
#pragma unroll
        for (int T = 0; T < 8; ++T) asm volatile("" : "+v"(cur[T]));      // the biases are consumed here (vae_wide16.hip.h)
        __syncthreads();                                    // also: the region is complete
        const float* __restrict__ xt = xw + q;
        f32x4 sc[4][2];
        float xv[8], bv[8];
        // operands of slab 0: inputs (transformed), first-half fragments
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xv[ks] = xt[4 * ks];
        if (SCALED) {
#pragma unroll
            for (int tb = 0; tb < 4; ++tb) {
                sc[tb][0] = *reinterpret_cast<const f32x4*>(side + q * 8 + tb * 32);
                sc[tb][1] = *reinterpret_cast<const f32x4*>(side + q * 8 + tb * 32 + 4);
            }
        }
#pragma unroll
        for (int T = 0; T < NT; ++T) av[T * 2] = *reinterpret_cast<const f32x4*>(wl + (T * 2) * 256);
        // (two columns per call.  Packed fp32 — v_pk_add/mul/fma_f32 on the pair — was tried: 40 -> 30 instructions per
        // slab, 2.42 -> 2.44 ms on C4; beside an MFMA stream a packed instruction costs more than the two it replaces,
        // MI355X_MICROARCH.md "price of one filler beside MFMAs")
        auto transform = [&](int kp) __attribute__((always_inline)) {      // columns 2kp and 2kp+1 of the slab
#pragma unroll
            for (int ks = 2 * kp; ks < 2 * kp + 2; ++ks)
                bv[ks] = SCALED ? robust_apply(xv[ks], sc[0][ks >> 2][ks & 3], sc[1][ks >> 2][ks & 3], sc[2][ks >> 2][ks & 3],
                                               sc[3][ks >> 2][ks & 3], clip) : xv[ks];
        };
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) transform(kp);
        VWS_MARK(2);
        int rb = 0;                                         // ring buffer of slab it
        const float* __restrict__ wp = w + 3 * VW16_SLAB;   // slab it+3
        const float* __restrict__ sp = ss + 3 * 32;
        // One memory or VALU operation goes behind every MFMA (a sched_barrier pins it there): an MFMA holds the pipe
        // for 32 cycles and issues in 4, so what the wave issues in between costs nothing, whereas the same operations
        // in a block before the MFMAs cost ~17 cycles each (measured: 1,665 cycles a half against 1,024).
        // The body has no branches: past the end of the layer it reads the slab before (operands nobody uses),
        // writes the ring buffer nobody reads and requests the last slab again.  With branches the compiler counts
        // the LDS operations of the longest path in every wait and carries the next slab's operands through copies.
        for (int it = 0; it < nin; ++it) {
            const int rn = rb == VWS_RING - 1 ? 0 : rb + 1;                 // ... of slab it+1
            const int rp = VWS_RING == 2 ? rb : (rn == VWS_RING - 1 ? 0 : rn + 1);   // ... of slab it+2
            VWS_STAMP(0);
            xt += it + 1 < nin ? 32 : 0;
            const float* __restrict__ wb = wl + rb * VW16_SLAB;
            const float* __restrict__ wn = wl + rn * VW16_SLAB;
            const float* __restrict__ sb = side + rn * VW16_SIDE + q * 8;
            float* const stp = stw + rp * VW16_SLAB;
            const bool more = it + 3 < nin;
            wp = more ? wp : w + (size_t)(nin - 1) * VW16_SLAB;
            sp = more ? sp : ss + (nin - 1) * 32;
            __builtin_amdgcn_sched_barrier(0);
            // first half; behind its MFMAs: the second-half fragments of this slab, the inputs and scaler entries of
            // the next one (their registers are free: this slab's inputs are transformed already)
            // (SPM side operations behind each of the half's 4 NT MFMAs: one for eight row tiles, two for four)
            auto side_a = [&](int n) __attribute__((always_inline)) {
                if (n < 8) { if (n < NT) av[n * 2 + 1] = *reinterpret_cast<const f32x4*>(wb + (n * 2 + 1) * 256); }
                else if (n < 16) xv[n - 8] = xt[4 * (n - 8)];
                else if (n < 24) { if (SCALED) sc[(n - 16) >> 1][(n - 16) & 1] = *reinterpret_cast<const f32x4*>(sb + ((n - 16) >> 1) * 32 + ((n - 16) & 1) * 4); }
            };
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    cur[T] = mfma16x4(av[T * 2][ks], bv[ks], cur[T]);
#pragma unroll
                    for (int j = 0; j < SPM; ++j) side_a((ks * NT + T) * SPM + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            VWS_STAMP(1);
            if (VWS_RING == 2) __syncthreads();             // every wave has read this slab's buffer: it takes slab it+2 below
            __builtin_amdgcn_sched_barrier(0);
            // second half; behind its MFMAs: the first-half fragments of the next slab (into the registers the first
            // half has left), slab it+2 into the ring (the buffer slab it-1 has left), slab it+3 requested, and the
            // next slab's inputs transformed, each into its register as soon as this slab has used it
            auto side_b = [&](int n) __attribute__((always_inline)) {
                if (n < 8) { if (n < NT) av[n * 2] = *reinterpret_cast<const f32x4*>(wn + (n * 2) * 256); }
                else if (n < 12) *reinterpret_cast<f32x4*>(stp + (n - 8) * 1024) = stg[n - 8];
                else if (n == 12) { if (side_thr) sdw[rp * VW16_SIDE] = side_next; }
                else if (n == 13) { vw16_stage_load(stg, wp, voff); if (side_thr) side_next = sp[0]; }
                else if (n == 16 || n == 17) transform(n - 16);     // (ks 0..3: the first half's)
                else if (n == 18) transform(2);             // (ks 4, 5: used by MFMAs 0..2 NT - 1 of this half, behind us here)
            };
#pragma unroll
            for (int ks = 4; ks < 8; ++ks)
#pragma unroll
                for (int T = 0; T < NT; ++T) {
                    cur[T] = mfma16x4(av[T * 2 + 1][ks & 3], bv[ks], cur[T]);
#pragma unroll
                    for (int j = 0; j < SPM; ++j) side_b(((ks - 4) * NT + T) * SPM + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            transform(3);                                   // (ks 6, 7: by MFMAs 16..31)
            wp += VW16_SLAB;
            sp += 32;
            __builtin_amdgcn_sched_barrier(0);
            VWS_STAMP(2);
            __syncthreads();
            VWS_STAMP(3);
            rb = rn;
        }
        VWS_MARK(3);
#pragma unroll
        for (int T = 0; T < 8; ++T) act4(cur[T], a.act[0]);
    }

    // ---- middle layers: registers -> registers; their weights through the idle ring (vae_wide16.hip.h).  The last
    // layer's first two slabs are requested before and land in registers meanwhile.
    const int nout_l = (a.dout[last] + 31) >> 5;
    vw16_stage_load(stg, b.w16[last], voff);
    if (nout_l > 1) vw16_stage_load(stg2, b.w16[last] + VW16_SLAB, voff);
    const bool mid_lds = b.mid_floats > 0;
    if (mid_lds) {
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(b.w16[1]);
        f32x4* dst = reinterpret_cast<f32x4*>(wt);
        const int n4 = b.mid_floats >> 2;
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) if ((int)threadIdx.x + u * 256 < n4) t[u] = src[threadIdx.x + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) if ((int)threadIdx.x + u * 256 < n4) dst[threadIdx.x + u * 256] = t[u];
        __syncthreads();
    }
    for (int p = 1; p < last; ++p) {
        const int nin = (a.din[p] + 15) >> 4, nout = (a.dout[p] + 15) >> 4;
        const float* __restrict__ w = mid_lds ? wt + (b.w16[p] - b.w16[1]) : b.w16[p];
        const float* __restrict__ bb = mid_lds ? wt + (b.b16[p] - b.w16[1]) : b.b16[p];
        // the reference's sampled z (myasp-A1.py:84, 94): the logvar head rides beside mu, from global memory (latent <= 32:
        // two output tiles at most, host-checked)
        const bool with_lv = b.wlv16 != nullptr && p == a.mu_idx;
        f32x4 nxt[8], lv[2];
#pragma unroll
        for (int ot = 0; ot < 8; ++ot)
            nxt[ot] = ot < nout ? *reinterpret_cast<const f32x4*>(bb + (ot * 4 + q) * 4) : f32x4{};
#pragma unroll
        for (int ot = 0; ot < 2; ++ot)
            lv[ot] = (with_lv && ot < nout) ? *reinterpret_cast<const f32x4*>(b.blv16 + (ot * 4 + q) * 4) : f32x4{};
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
                if (with_lv) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int ot = 0; ot < 2; ++ot)
                            if (ot < nout)
                                lv[ot] = mfma16x4(b.wlv16[(((size_t)ot * nin + Tin) * 4 + r) * 64 + lane], cur[Tin][r], lv[ot]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ot = 0; ot < 8; ++ot) { cur[ot] = nxt[ot]; act4(cur[ot], a.act[p]); }
        if (with_lv) {
            // z = mu + exp(0.5 * clip(logvar)) * eps, vae_kernel's operations (kernels.hip.h); rows 16 ot + 4 q + r of
            // this lane's window (a lane without a window reads the draws of its wave's last one and stores nothing)
            const float* __restrict__ ep = a.eps + (a.seg.win_off[s] + k0 + 16 * wave + jc) * (long long)a.dout[p];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * ot + 4 * q + r;
                    if (ot < nout && d < a.dout[p]) {
                        float l = lv[ot][r];
                        if (a.lv_clip > 0.0f) l = fminf(fmaxf(l, -a.lv_clip), a.lv_clip);
                        cur[ot][r] += __builtin_amdgcn_exp2f(0.72134752044448170f * l) * ep[d];
                    }
                }
        }
    }
    if (mid_lds) __syncthreads();                           // the ring is rewritten below
    VWS_MARK(4);

    // ---- last layer: hidden -> D, weights streamed per 32-element output slab, error fused
    float sq = 0.0f;
    {
        const int p = last;
        const int nout = nout_l;
        const float* __restrict__ w = b.w16[p];
        const int act = a.act[p];
        // side tile, last layer: the scaler tables in natural column order, then the slab's 32 biases
        const float* __restrict__ ss = (SCALED && threadIdx.x < 128) ? a.rs + (size_t)(threadIdx.x >> 5) * a.rsP + (threadIdx.x & 31)
                                       : (threadIdx.x >= 128 && threadIdx.x < 160) ? b.b16[p] + (threadIdx.x - 128) : w;
        const bool side_thr = (SCALED && threadIdx.x < 128) || (threadIdx.x >= 128 && threadIdx.x < 160);
        float side_next = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(stw + i * 1024) = stg[i];
        if (side_thr) sdw[0] = ss[0];
        if (nout > 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(stw + VW16_SLAB + i * 1024) = stg2[i];
            if (side_thr) sdw[VW16_SIDE] = ss[32];
        }
        {
            const int s2 = nout > 2 ? 2 : nout - 1;
            vw16_stage_load(stg, w + (size_t)s2 * VW16_SLAB, voff);
            if (side_thr) side_next = ss[s2 * 32];
        }
        __syncthreads();
        const float* __restrict__ xt = xw + 4 * q;
        f32x4 sc[2][4], accn0, accn1;
        float xv[8], xt8[8];
        // operands of slab 0: biases, inputs (transformed), first-half fragments (input tiles 0..3)
        {
            const float* __restrict__ sb = side + 4 * q;
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
            for (int Tin = 0; Tin < HT; ++Tin) {
                av[Tin] = *reinterpret_cast<const f32x4*>(wl + Tin * 256);
                av[8 + Tin] = *reinterpret_cast<const f32x4*>(wl + (8 + Tin) * 256);
            }
        }
        auto transform = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xt8[T2 * 4 + r] = SCALED ? robust_apply(xv[T2 * 4 + r], sc[T2][0][r], sc[T2][1][r], sc[T2][2][r], sc[T2][3][r], clip)
                                             : xv[T2 * 4 + r];
        };
        transform();
        VWS_MARK(5);
        int rb = 0;
        const float* __restrict__ wp = w + 3 * VW16_SLAB;   // slab ot+3
        const float* __restrict__ sp = ss + 3 * 32;
        for (int ot = 0; ot < nout; ++ot) {                 // (one operation behind every MFMA, no branches: see layer 0)
            const int rn = rb == VWS_RING - 1 ? 0 : rb + 1;
            const int rp = VWS_RING == 2 ? rb : (rn == VWS_RING - 1 ? 0 : rn + 1);
            f32x4 acc0 = accn0, acc1 = accn1;
            xt += ot + 1 < nout ? 32 : 0;
            const float* __restrict__ wb = wl + rb * VW16_SLAB;
            const float* __restrict__ wn = wl + rn * VW16_SLAB;
            const float* __restrict__ sb = side + rn * VW16_SIDE + 4 * q;
            float* const stp = stw + rp * VW16_SLAB;
            const bool more = ot + 3 < nout;
            wp = more ? wp : w + (size_t)(nout - 1) * VW16_SLAB;
            sp = more ? sp : ss + (nout - 1) * 32;
            __builtin_amdgcn_sched_barrier(0);
            // first half (input tiles 0..3); behind its MFMAs: the second-half fragments of this slab, the biases,
            // inputs and scaler entries of the next one
            // (input tiles 0..HT-1 in this half, HT..NT-1 in the second; SPM side operations behind each MFMA)
            auto side_c0 = [&](int n) __attribute__((always_inline)) {
                if (n < 4) { if (n < HT) av[HT + n] = *reinterpret_cast<const f32x4*>(wb + (HT + n) * 256); }
                else if (n == 4) accn0 = *reinterpret_cast<const f32x4*>(sb + 128);
                else if (n < 9) { if (SCALED) sc[0][n - 5] = *reinterpret_cast<const f32x4*>(sb + (n - 5) * 32); }
                else if (n < 13) xv[n - 9] = xt[n - 9];
            };
            auto side_c1 = [&](int n) __attribute__((always_inline)) {
                if (n < 4) { if (n < HT) av[8 + HT + n] = *reinterpret_cast<const f32x4*>(wb + (8 + HT + n) * 256); }
                else if (n == 4) accn1 = *reinterpret_cast<const f32x4*>(sb + 128 + 16);
                else if (n < 9) { if (SCALED) sc[1][n - 5] = *reinterpret_cast<const f32x4*>(sb + (n - 5) * 32 + 16); }
                else if (n < 13) xv[4 + n - 9] = xt[16 + n - 9];
            };
#pragma unroll
            for (int Tin = 0; Tin < HT; ++Tin)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0 = mfma16x4(av[Tin][r], cur[Tin][r], acc0);
#pragma unroll
                    for (int j = 0; j < SPM; ++j) side_c0((Tin * 4 + r) * SPM + j);
                    __builtin_amdgcn_sched_barrier(0);
                    acc1 = mfma16x4(av[8 + Tin][r], cur[Tin][r], acc1);
#pragma unroll
                    for (int j = 0; j < SPM; ++j) side_c1((Tin * 4 + r) * SPM + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            if (VWS_RING == 2) __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            // second half (input tiles 4..7); behind its MFMAs: the first-half fragments of the next slab, slab ot+2
            // into the ring, slab ot+3 requested
            auto side_d0 = [&](int n) __attribute__((always_inline)) {
                if (n < 4) { if (n < HT) av[n] = *reinterpret_cast<const f32x4*>(wn + n * 256); }
                else if (n < 8) *reinterpret_cast<f32x4*>(stp + (n - 4) * 1024) = stg[n - 4];
                else if (n == 8) { vw16_stage_load(stg, wp, voff); }
            };
            auto side_d1 = [&](int n) __attribute__((always_inline)) {
                if (n < 4) { if (n < HT) av[8 + n] = *reinterpret_cast<const f32x4*>(wn + (8 + n) * 256); }
                else if (n == 4) { if (side_thr) sdw[rp * VW16_SIDE] = side_next; }
                else if (n == 8) { if (side_thr) side_next = sp[0]; }
            };
#pragma unroll
            for (int Tin = HT; Tin < NT; ++Tin)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0 = mfma16x4(av[Tin][r], cur[Tin][r], acc0);
#pragma unroll
                    for (int j = 0; j < SPM; ++j) side_d0(((Tin - HT) * 4 + r) * SPM + j);
                    __builtin_amdgcn_sched_barrier(0);
                    acc1 = mfma16x4(av[8 + Tin][r], cur[Tin][r], acc1);
#pragma unroll
                    for (int j = 0; j < SPM; ++j) side_d1(((Tin - HT) * 4 + r) * SPM + j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            wp += VW16_SLAB;
            sp += 32;
            // epilogue: activation, difference to the transformed input, running sum of squares (vw16_epilogue's
            // operations in its order); then the next slab's inputs transformed
            sq = vwp_ll_epi(acc0, acc1, xt8, act, ot + 1 == nout, ot * 32 + 4 * q, D, sq);
            transform();
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            rb = rn;
        }
    }
    VWS_MARK(6);
    sq += __shfl_xor(sq, 16);                               // the four q of a window hold different rows
    sq += __shfl_xor(sq, 32);
    if (a.err && q == 0 && col < count) a.err[wout] = sq / (float)D;
}
