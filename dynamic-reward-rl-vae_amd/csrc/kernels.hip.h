// CDNA4 (gfx950) kernels of the window-scoring path.  Wave = 64 lanes.
//
// Everything GEMM-shaped runs on v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain) in
// the "transposed" orientation D[row = output feature][col = window]:
//   A operand (1 VGPR): lane l holds A[row = l&31][k = l>>5]   -> a weight, pre-packed on the host
//   B operand (1 VGPR): lane l holds B[k = l>>5][col = l&31]   -> an activation of window l&31
//   C/D (16 VGPRs):     lane l, reg r holds D[row = rowmap(r, l>>5)][col = l&31],
//                       rowmap(r, h) = (r&3) + 8*(r>>2) + 4*h
// With this orientation an accumulator register *is* a valid B operand of the next product
// (k-step r of an input tile pairs rows rowmap(r,0) and rowmap(r,1)), so activations chain from
// layer to layer, and h_t feeds step t+1, without leaving the register file; the weights are
// packed on the host in exactly that k order (pack_* in drlvae.hip).
//
// Kernels, in file order (DESIGN.md section 4 has the table with shapes and bounds):
//   vae_kernel<MAXT>            VAE chain, narrow windows / explicit batches / eps path
//   (vae_wide16.hip.h)          vae_wide16_kernel<SCALED>: VAE, wide windows — LDS-staged windows, weights through an LDS ring
//   lstm_generic_kernel<NT>     fallback LSTM (whole cell per wave, weights from L2)
//   xproj_kernel<NT>            input projections once per sample
//   lstm_rec_kernel<NT,FLAGGED> recurrence from the projections, H <= 64
//   lstm_rec2_kernel<NT>        the same for two candidate states per window, both in one wave
//   lstm_h128i2_kernel<NG>      univariate H <= 128: U stationary in registers, persistent grid
//   lstm_h128p_kernel<NG>       the same fed from projections (any input width)
//   lstm_gemv_kernel<NT>        estimator.predict at batch 1..4 (VALU matrix-vector)
//   finalize_kernel, rollout_*_kernel, al_*_kernel   per-window epilogues
// lstm_w16.hip.h adds lstm_h128i2_w16_kernel: the univariate recurrence on 16-window tiles
// (v_mfma_f32_16x16x4_f32) for calls too small to fill the chip with 32-window tiles.
// lstm_f16x3.hip.h adds the opt-in 3-term f16-split variant of lstm_h128i2_kernel.
// VW16_ABL / H128_ABL / REC_ABL are timing-only ablation switches (wrong results) used to price the parts of a
// kernel; the default build has them off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DRL_MAX_VAE_LAYERS 8
#define DRL_VAE_MAX_TILES 4   // hidden widths <= 128

__device__ __forceinline__ int rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Pins a stationary weight into the accumulator half of the register file.  hipcc otherwise
// treats AGPRs as spill space and copies every such weight back with v_accvgpr_read in front of
// its MFMA, and on gfx950 every VALU instruction adds to the fp32 MFMA time (they do not overlap:
// tools/ubench_mfma_valu.hip).  A value *defined* in register class "a" stays there, and the MFMA
// builtin takes an AGPR as its A operand directly (AV operand), with the compiler still in charge
// of every hazard.
__device__ __forceinline__ float pin_agpr(float x) {
    float r;
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(x));
    return r;
}

// sigmoid / tanh through v_exp_f32 + v_rcp_f32 (both 1 ulp): absolute error ~1e-7.
__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}

// ------------------------------------------------------------------------------------------------
// Series segments.  Windows never cross a series; a "tile" is a run of consecutive windows of one
// series (32 for the per-wave kernels, 64 for the h128i2 workgroup kernel).
struct SegTab {
    const long long* samp_off;  // [n_seg+1] first sample of each series in the concatenated input
    const long long* win_off;   // [n_seg+1] first output window of each series
    const int* tile_off;        // [n_seg+1] first tile of each series
    const float* in_scale;      // [n_seg] or null
    const float* in_offset;     // [n_seg] or null
    int n_seg;
    int W;
};

__device__ __forceinline__ int find_seg(const int* __restrict__ tile_off, int n_seg, int tile) {
    int lo = 0, hi = n_seg;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (tile_off[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

// Persistent kernels look a tile's series up once per tile; out of global memory that is a chain of
// ~log2(n_seg)+2 dependent loads (microseconds).  Up to SEG_LDS series the tables are copied to LDS
// once per workgroup instead.
#define SEG_LDS 512
struct SegLds {
    long long samp[SEG_LDS + 1];
    long long win[SEG_LDS + 1];
    int toff[SEG_LDS + 1];
};
struct SegHit { int s; long long samp0, nsamp, win0, Ks; int tile0; };
__device__ __forceinline__ void seg_stage(SegLds& d, const SegTab& seg) {
    if (seg.n_seg > SEG_LDS) return;
    for (int i = threadIdx.x; i <= seg.n_seg; i += blockDim.x) {
        d.samp[i] = seg.samp_off[i]; d.win[i] = seg.win_off[i]; d.toff[i] = seg.tile_off[i];
    }
}
__device__ __forceinline__ SegHit seg_lookup(const SegLds& d, const SegTab& seg, int tile) {
    SegHit h;
    if (seg.n_seg <= SEG_LDS) {
        h.s = find_seg(d.toff, seg.n_seg, tile);
        h.samp0 = d.samp[h.s]; h.nsamp = d.samp[h.s + 1] - h.samp0;
        h.win0 = d.win[h.s]; h.Ks = d.win[h.s + 1] - h.win0; h.tile0 = d.toff[h.s];
    } else {
        h.s = find_seg(seg.tile_off, seg.n_seg, tile);
        h.samp0 = seg.samp_off[h.s]; h.nsamp = seg.samp_off[h.s + 1] - h.samp0;
        h.win0 = seg.win_off[h.s]; h.Ks = seg.win_off[h.s + 1] - h.win0; h.tile0 = seg.tile_off[h.s];
    }
    return h;
}

// reference env.py:78-79 (sklearn MinMaxScaler): x*scale_ + min_ with two float32 roundings
__device__ __forceinline__ float minmax_apply(float x, float sc, float of) {
    float m = x * sc;
    asm volatile("" : "+v"(m));        // keep the product rounded: no fma contraction
    return m + of;
}

// reference myasp-wadi.py:138-139 (sklearn RobustScaler.transform on the float32 windows, then np.clip):
//   X -= center_   numpy rounds x - center_ to float32 once,
//   X /= scale_    then t / scale_ (float64 scale_) to float32 once.
// center_ and 1/scale_ arrive as float32 hi+lo pairs of the caller's float64 parameters.  x - c_hi is
// exact whenever the two are within a factor of two of each other (Sterbenz) — the cancelling case, where
// a center truncated to float32 would cost |c_lo|/scale_ — so (x - c_hi) - c_lo is the float32 rounding
// of x - center_ there, and within an ulp of it elsewhere; t*inv_hi + t*inv_lo with one fma is the
// float32 rounding of t/scale_ up to a 2^-48 relative term.  ONE formula for every kernel.
__device__ __forceinline__ float robust_apply(float x, float c_hi, float c_lo, float i_hi, float i_lo, float clip) {
    float t = x - c_hi;
    t = t - c_lo;
    const float p = t * i_lo;
    return __builtin_amdgcn_fmed3f(fmaf(t, i_hi, p), -clip, clip);
}

// ------------------------------------------------------------------------------------------------
// DRL_CLK_STAMPS (diagnostic builds only, tools/kernel_clock.py): workgroup 0 stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) at kernel entry and exit; their quotient is the clock the chip holds under the kernel's load
// (MI355X_MICROARCH.md, "DVFS give-back", item 6).  In the shipped library nothing is stamped.
#ifdef DRL_CLK_STAMPS
__device__ unsigned long long drl_clk_dbg[4];
struct ClkStamp {
    bool on;
    __device__ ClkStamp() : on(blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        if (on) { drl_clk_dbg[0] = __builtin_amdgcn_s_memtime(); drl_clk_dbg[1] = __builtin_amdgcn_s_memrealtime(); }
    }
    __device__ ~ClkStamp() {
        if (on) { drl_clk_dbg[2] = __builtin_amdgcn_s_memtime(); drl_clk_dbg[3] = __builtin_amdgcn_s_memrealtime(); }
    }
};
#define DRL_CLK ClkStamp drl_clk_stamp_;
#else
#define DRL_CLK
#endif

// K2: VAE mean path + reconstruction error.  One wave = 32 windows, no LDS, no barriers.
struct VaeArgs {
    // path = encoder hidden..., mu, decoder...  (the logvar head rides beside mu when eps != null)
    int n;                                   // layers on the path (>= 2)
    int din[DRL_MAX_VAE_LAYERS];
    int dout[DRL_MAX_VAE_LAYERS];
    int act[DRL_MAX_VAE_LAYERS];
    const float* w[DRL_MAX_VAE_LAYERS];      // packed [out_tile][in_tile][16][64]
    const float* b[DRL_MAX_VAE_LAYERS];      // packed [out_tile][2][16]
    int mu_idx;                              // position of mu on the path
    const float* w_lv;                       // packed like w[mu_idx]
    const float* b_lv;
    float lv_clip;                           // <= 0: none
    // input
    const float* x;                          // flat series [samples*F] or explicit batch [B*D]
    long long win_stride;                    // elements between consecutive windows (F, or D)
    int lag_elems;                           // element offset of window 0 inside its series: (1-lag)*F
    int D;                                   // window length in elements (W*F)
    // RobustScaler (reference myasp-wadi.py:138): per flattened column, padded to P = a multiple of 32
    // columns, [4][P] = c_hi | c_lo | inv_hi | inv_lo with center_ = c_hi + c_lo and 1/scale_ = inv_hi +
    // inv_lo split from the caller's float64 parameters on the host (pad: 0 | 0 | 1 | 0); null = none
    const float* rs;
    int rsP;
    float clip;                              // <= 0: none
    const float* eps;                        // [K_total, latent] or null
    SegTab seg;
    int n_tiles;
    // outputs
    float* err;                              // [K_total] or null
    float* recon;                            // [K_total, D] or null (drl_vae_predict)
    // the whole weight pack (every w/b pointer above points into it): vae_kernel<.., true> copies it to LDS
    const float* pack;
    int pack_floats;                         // a multiple of 16
};

// act 3 = sigmoid whose layer was packed pre-scaled by -log2(e): 1/(1+2^y)
__device__ __forceinline__ float apply_act(float y, int act) {
    if (act == 1) return fmaxf(y, 0.0f);
    if (act == 2) return fast_sigmoid(y);
    if (act == 3) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y));
    return y;
}

// Activation of a whole accumulator tile with the (wave-uniform, run-time) code tested ONCE: a
// per-element apply_act(y, act) leaves a chain of scalar branches around every element, which
// serialises the transcendental latencies.
template <int ACT>
__device__ __forceinline__ void act16_ct(f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], ACT);
}
__device__ __forceinline__ void act16(f32x16& v, int act) {
    if (act == 1) act16_ct<1>(v);
    else if (act == 2) act16_ct<2>(v);
    else if (act == 3) act16_ct<3>(v);
}

__device__ __forceinline__ f32x16 load_bias16(const float* __restrict__ bp, int tile, int half) {
    const f32x4* p = reinterpret_cast<const f32x4*>(bp + (tile * 2 + half) * 16);
    f32x4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 v;
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    v[8] = c[0]; v[9] = c[1]; v[10] = c[2]; v[11] = c[3];
    v[12] = d[0]; v[13] = d[1]; v[14] = d[2]; v[15] = d[3];
    return v;
}

// The 16 inputs one lane feeds to the MFMAs of input tile `it` (features it*32 + rowmap(r, half)).
// All loads are issued before the first use (clamped indices instead of a branch per element: one
// round trip to memory per tile, not sixteen).
__device__ __forceinline__ void vae_input16(const VaeArgs& a, const float* __restrict__ xin, int it, int half,
                                            bool has_norm, float sc, float of, float (&bv)[16]) {
    int dc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = it * 32 + rowmap(r, half);
        dc[r] = d < a.D ? d : a.D - 1;
        bv[r] = xin[dc[r]];
    }
    if (has_norm) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = minmax_apply(bv[r], sc, of);
    }
    if (a.rs) {
        const float clip = a.clip > 0.0f ? a.clip : 3.0e38f;
        float ch[16], cl[16], ih[16], il[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ch[r] = a.rs[dc[r]]; cl[r] = a.rs[a.rsP + dc[r]];
            ih[r] = a.rs[2 * a.rsP + dc[r]]; il[r] = a.rs[3 * a.rsP + dc[r]];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = robust_apply(bv[r], ch[r], cl[r], ih[r], il[r], clip);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = (it * 32 + rowmap(r, half)) < a.D ? bv[r] : 0.0f;
}

// MAXT = accumulator tiles kept per layer: 2 when every hidden width is <= 64 (the univariate
// VAEs), else MAXT.  The kernel is latency-bound (weights come from L2 per MFMA, 256
// MFMAs per 32 windows), so the narrow instantiation's lower register count buys resident waves.
// LDSW: calls of a few tiles (vae.predict on one window, one short series) have nothing to hide that
// latency with — one wave waits ~1-2 us per group of weight loads, 40 us for the chain.  There the
// workgroup first copies the whole weight pack (70 KB for the univariate VAEs) to LDS, all loads in
// flight at once, and the chain reads its A fragments from LDS.  Same MFMAs in the same order.
template <int MAXT, bool LDSW = false>
__global__ __launch_bounds__(256, (LDSW ? 1 : (MAXT <= 2 ? 3 : 1))) void vae_kernel(VaeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float vk_pack[];
    if constexpr (LDSW) {
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.pack);
        f32x4* dst = reinterpret_cast<f32x4*>(vk_pack);
        const int n4 = a.pack_floats >> 2;
        for (int i0 = threadIdx.x; i0 < n4; i0 += 256 * 8) {
            f32x4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * 256 < n4) t[u] = src[i0 + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * 256 < n4) dst[i0 + u * 256] = t[u];
        }
        __syncthreads();
    }
    auto wp = [&](const float* g) -> const float* {
        if constexpr (LDSW) return vk_pack + (g - a.pack); else return g;
    };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.n_tiles) return;                       // wave-uniform
    const int s = find_seg(a.seg.tile_off, a.seg.n_seg, tile);
    const long long samp0 = a.seg.samp_off[s];
    const long long nsamp = a.seg.samp_off[s + 1] - samp0;
    const long long Ks = a.seg.win_off[s + 1] - a.seg.win_off[s];
    const long long k0 = (long long)(tile - a.seg.tile_off[s]) * 32;
    const int count = (int)(Ks - k0 < 32 ? Ks - k0 : 32);
    const int jj = j < count ? j : count - 1;
    (void)nsamp;
    const bool has_norm = a.seg.in_scale != nullptr;
    const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
    const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
    const float* __restrict__ xin = a.x + samp0 * a.win_stride + a.lag_elems + (k0 + jj) * a.win_stride;
    const long long wout = a.seg.win_off[s] + k0 + jj;

    f32x16 cur[MAXT], nxt[MAXT];

    // ---- layer 0: input from memory
    {
        const int nin = (a.din[0] + 31) >> 5, nout = (a.dout[0] + 31) >> 5;
        const float* __restrict__ w = wp(a.w[0]);
#pragma unroll
        for (int ot = 0; ot < MAXT; ++ot)
            cur[ot] = ot < nout ? load_bias16(wp(a.b[0]), ot, half) : f32x16{};
        if constexpr (MAXT > 2) {
            // wide layers (explicit batches of wide windows: nin up to ~100 tiles, one wave alone with the L2
            // latency): the A fragments and inputs of tile it+1 are requested before the MFMAs of tile it
            float wa[MAXT * 16], bv[16];
            auto load_tile = [&](int it, float (&wt)[MAXT * 16], float (&bt)[16]) {
#pragma unroll
                for (int ot = 0; ot < MAXT; ++ot)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        wt[ot * 16 + r] = w[(((size_t)it * nout + (ot < nout ? ot : nout - 1)) * 16 + r) * 64 + lane];
                vae_input16(a, xin, it, half, has_norm, sc, of, bt);
            };
            load_tile(0, wa, bv);
            for (int it = 0; it < nin; ++it) {
                float wn[MAXT * 16], bn[16];
                load_tile(it + 1 < nin ? it + 1 : it, wn, bn);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int ot = 0; ot < MAXT; ++ot)
                        if (ot < nout) cur[ot] = mfma32(wa[ot * 16 + r], bv[r], cur[ot]);
                }
#pragma unroll
                for (int q = 0; q < MAXT * 16; ++q) wa[q] = wn[q];
#pragma unroll
                for (int q = 0; q < 16; ++q) bv[q] = bn[q];
            }
        } else {
            for (int it = 0; it < nin; ++it) {
                float bv[16];
                vae_input16(a, xin, it, half, has_norm, sc, of, bv);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int ot = 0; ot < MAXT; ++ot)
                        if (ot < nout)
                            cur[ot] = mfma32(w[(((size_t)it * nout + ot) * 16 + r) * 64 + lane], bv[r], cur[ot]);
                }
            }
        }
#pragma unroll
        for (int ot = 0; ot < MAXT; ++ot) act16(cur[ot], a.act[0]);
    }

    // ---- middle layers: registers -> registers
    for (int p = 1; p < a.n - 1; ++p) {
        const int nin = (a.din[p] + 31) >> 5, nout = (a.dout[p] + 31) >> 5;
        const float* __restrict__ w = wp(a.w[p]);
        const float* __restrict__ wlv = wp(a.w_lv);
        const bool with_lv = (p == a.mu_idx) && a.eps != nullptr;
        f32x16 lv = {};
#pragma unroll
        for (int ot = 0; ot < MAXT; ++ot)
            nxt[ot] = ot < nout ? load_bias16(wp(a.b[p]), ot, half) : f32x16{};
        if (with_lv) lv = load_bias16(wp(a.b_lv), 0, half);
#pragma unroll
        for (int it = 0; it < MAXT; ++it) {
            if (it < nin) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int ot = 0; ot < MAXT; ++ot)
                        if (ot < nout)
                            nxt[ot] = mfma32(w[(((size_t)ot * nin + it) * 16 + r) * 64 + lane], cur[it][r], nxt[ot]);
                    if (with_lv)   // latent <= 32: one output tile
                        lv = mfma32(wlv[(((size_t)it) * 16 + r) * 64 + lane], cur[it][r], lv);
                }
            }
        }
#pragma unroll
        for (int ot = 0; ot < MAXT; ++ot) { cur[ot] = nxt[ot]; act16(cur[ot], a.act[p]); }
        if (with_lv) {
            // z = mu + exp(0.5*clip(logvar))*eps   (reference myasp-A1.py:84, 94)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = rowmap(r, half);
                if (d < a.dout[p]) {
                    float l = lv[r];
                    if (a.lv_clip > 0.0f) l = fminf(fmaxf(l, -a.lv_clip), a.lv_clip);
                    // exp(l/2) as 2^(l * log2(e)/2): the one-instruction form every other exponential here takes
                    cur[0][r] += __builtin_amdgcn_exp2f(0.72134752044448170f * l) * a.eps[wout * a.dout[p] + d];
                }
            }
        }
    }

    // ---- last layer: registers -> sigmoid -> squared error against the input
    {
        const int p = a.n - 1;
        const int nin = (a.din[p] + 31) >> 5, nout = (a.dout[p] + 31) >> 5;
        const float* __restrict__ w = wp(a.w[p]);
        float sq = 0.0f;
        float wa[MAXT > 2 ? MAXT * 16 : 1], wn[MAXT > 2 ? MAXT * 16 : 1];
        auto load_out_tile = [&](int ot, float (&wt)[MAXT > 2 ? MAXT * 16 : 1]) {
            if constexpr (MAXT > 2) {
#pragma unroll
                for (int it = 0; it < MAXT; ++it)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        wt[it * 16 + r] = w[(((size_t)ot * nin + (it < nin ? it : nin - 1)) * 16 + r) * 64 + lane];
            }
        };
        load_out_tile(0, wa);
        for (int ot = 0; ot < nout; ++ot) {
            f32x16 acc = load_bias16(wp(a.b[p]), ot, half);
            float xv16[16];                              // requested before the MFMAs that hide them
            vae_input16(a, xin, ot, half, has_norm, sc, of, xv16);
            load_out_tile(ot + 1 < nout ? ot + 1 : ot, wn);      // wide layers: next tile's fragments in flight
#pragma unroll
            for (int it = 0; it < MAXT; ++it) {
                if (it < nin) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if constexpr (MAXT > 2) acc = mfma32(wa[it * 16 + r], cur[it][r], acc);
                        else acc = mfma32(w[(((size_t)ot * nin + it) * 16 + r) * 64 + lane], cur[it][r], acc);
                    }
                }
            }
            if constexpr (MAXT > 2) {
#pragma unroll
                for (int q = 0; q < MAXT * 16; ++q) wa[q] = wn[q];
            }
            act16(acc, a.act[p]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = ot * 32 + rowmap(r, half);
                const float y = acc[r];
                const float xv = xv16[r];
                if (d < a.D) {
                    const float df = y - xv;
                    sq = fmaf(df, df, sq);
                    if (a.recon && j < count) a.recon[wout * a.D + d] = y;
                }
            }
        }
        sq += __shfl_xor(sq, 32);                      // the two halves hold different rows
        if (a.err && half == 0 && j < count) a.err[wout] = sq / (float)a.D;
    }
}


// ------------------------------------------------------------------------------------------------
// K3: LSTM-DQN.
struct LstmArgs {
    int I, H, A, W;
    float forget_bias;
    const float* wx;        // packed [ceil(I/2)][NT][64]
    const float* u;         // packed [H/2][NT][64]
    const float* bias;      // packed [NT][2][16]
    const float* w_out;     // [H][A]
    const float* b_out;     // [A]
    // input
    const float* x;         // series [samples, F] or explicit states [B, W, I]
    int F;                  // features taken from x per row (I-1 when the flag is synthesised, I when explicit states)
    long long win_stride;   // elements between windows (F, or W*I)
    long long row_stride;   // elements between rows of one window (F, or I)
    int row0;               // first row of window 0 relative to the series start (1, or 0)
    int flag_mode;          // drl_flag_mode; -1 = no synthesised flag (explicit states)
    const float* flags;     // per-sample flags (explicit mode)
    SegTab seg;
    int n_tiles;
    int S;                  // candidate states per window (gridDim.y)
    float* q;               // [K_total, S, A]
    const int* tile_list;   // null, or the n_tiles tile indices to work on (the sparse passes of drl_rollout_batch)
    int tile0;              // lstm_h128i2_kernel: this launch covers tiles [tile0, n_tiles) (0 unless the call is split)
    // argmax epilogue (reference myasp-A1.py:227 np.argmax; :255 AL margin; myasp-wadi.py:328 WaDi rule): when the
    // kernel that computes a window's Q holds all the values the rule needs, it writes action and margin itself
    // (host: fused_epilogue); otherwise finalize_kernel does, from q.  Either may be null.
    int* action;            // [K_total]
    float* margin;          // [K_total]
    int rule;               // drl_action_rule
};

// np.argmax over the actions of one state, ties -> the first, and |q0 - q1|: the operations of finalize_kernel
struct ActEpi {
    float q0, q1, bv;
    int best;
    __device__ __forceinline__ void push(int act, float v) {
        if (act == 0) { q0 = v; bv = v; best = 0; }
        else {
            if (act == 1) q1 = v;
            if (v > bv) { bv = v; best = act; }
        }
    }
    __device__ __forceinline__ void store(const LstmArgs& a, long long win) const {
        if (a.action) a.action[win] = best;
        if (a.margin) a.margin[win] = fabsf(q0 - q1);
    }
};

__device__ __forceinline__ float synth_flag(int mode, long long k, int t, int W, int cand,
                                            const float* __restrict__ flags, long long sample) {
    if (mode == 1) return (k < W && t == W - 1 - (int)k) ? 1.0f : 0.0f;
    if (mode == 2) return flags[sample];
    if (mode == 3) return (float)cand;
    return 0.0f;
}

// One LSTM cell update for the four gate registers of one (unit, window).  The gate
// pre-activations arrive pre-scaled (pack_lstm in drlvae.hip multiplies the i|f|o columns by
// -log2(e), the j columns by -2*log2(e), and folds forget_bias into the f bias), so each gate is
// one v_exp_f32 + one v_rcp_f32:  sigmoid(z) = 1/(1+2^z'),  tanh(z) = 2/(1+2^z') - 1.
__device__ __forceinline__ float lstm_cell(float zi, float zj, float zf, float zo, float fb, float& c) {
    (void)fb;
    const float si = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zi));
    const float tj = fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zj)), -1.0f);
    const float sf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zf));
    const float so = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zo));
    c = fmaf(sf, c, si * tj);
    return so * fast_tanh(c);
}

// Generic shapes: one wave = 32 windows of one candidate state, weights streamed from L2 in
// packed A-fragment order (256 B per MFMA, coalesced).  c stays in registers; h_t is parked in a
// wave-private LDS slab in B-fragment order (each lane re-reads exactly what it wrote, so no
// barrier is needed) so that the k loop can stay a rolled loop.
// NT = gate tiles (tile tau, row i: unit 8*tau + (i>>2), gate i&3); H <= 8*NT, zero padded.
template <int NT>
__global__ __launch_bounds__(256, 1) void lstm_generic_kernel(LstmArgs a) {
    __shared__ __attribute__((aligned(16))) float s_h[4][NT * 4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.n_tiles) return;
    const int cand = blockIdx.y;
    const int s = find_seg(a.seg.tile_off, a.seg.n_seg, tile);
    const long long samp0 = a.seg.samp_off[s];
    const long long Ks = a.seg.win_off[s + 1] - a.seg.win_off[s];
    const long long k0 = (long long)(tile - a.seg.tile_off[s]) * 32;
    const int count = (int)(Ks - k0 < 32 ? Ks - k0 : 32);
    const int jj = j < count ? j : count - 1;
    const long long k = k0 + jj;
    const bool has_norm = a.seg.in_scale != nullptr;
    const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
    const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
    const float* __restrict__ xw = a.x + samp0 * a.row_stride + (long long)a.row0 * a.row_stride + k * a.win_stride;
    const long long flag_base = samp0 + a.row0 + k;      // sample index of row 0 (explicit flags)
    const int nks = (a.I + 1) >> 1;
    float* __restrict__ hw = &s_h[wave][0][lane];

    f32x16 acc[NT];
    float c[NT][4];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) c[tl][qd] = 0.0f;

    for (int t = 0; t < a.W; ++t) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = load_bias16(a.bias, tl, half);
        // input part: [x_t, flag] . Wx
        const float* __restrict__ xr = xw + (long long)t * a.row_stride;
#pragma unroll 2
        for (int ks = 0; ks < nks; ++ks) {
            const int feat = 2 * ks + half;
            float bv = 0.0f;
            if (feat < a.F) {
                bv = xr[feat];
                if (has_norm) bv = minmax_apply(bv, sc, of);
            } else if (feat == a.F && a.flag_mode >= 0) {
                bv = synth_flag(a.flag_mode, k, t, a.W, cand, a.flags, flag_base + t);
            }
            const float* __restrict__ wp = a.wx + ((size_t)ks * NT) * 64 + lane;
#pragma unroll
            for (int tl = 0; tl < NT; ++tl) acc[tl] = mfma32(wp[tl * 64], bv, acc[tl]);
        }
        // recurrent part: h_{t-1} . U
        if (t > 0) {
#pragma unroll 2
            for (int kk = 0; kk < NT * 4; ++kk) {
                const float hv = hw[kk * 64];
                const float* __restrict__ up = a.u + ((size_t)kk * NT) * 64 + lane;
#pragma unroll
                for (int tl = 0; tl < NT; ++tl) acc[tl] = mfma32(up[tl * 64], hv, acc[tl]);
            }
        }
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                hw[(tl * 4 + qd) * 64] = lstm_cell(acc[tl][4 * qd], acc[tl][4 * qd + 1], acc[tl][4 * qd + 2],
                                                   acc[tl][4 * qd + 3], a.forget_bias, c[tl][qd]);
            __builtin_amdgcn_sched_barrier(0);          // keep live ranges short: one tile at a time
        }
    }

    // Q head: lane holds units 8*tl + 2*qd + half of window j
    ActEpi epi;
    for (int act = 0; act < a.A; ++act) {
        float part = 0.0f;
        for (int kk = 0; kk < NT * 4; ++kk) {
            const int unit = 8 * (kk >> 2) + 2 * (kk & 3) + half;
            if (unit < a.H) part = fmaf(hw[kk * 64], a.w_out[unit * a.A + act], part);
        }
        part += __shfl_xor(part, 32);
        const float qv = part + a.b_out[act];
        epi.push(act, qv);
        if (half == 0 && j < count)
            a.q[((a.seg.win_off[s] + k0 + j) * a.S + cand) * a.A + act] = qv;
    }
    if (a.S == 1 && half == 0 && j < count) epi.store(a, a.seg.win_off[s] + k0 + j);
}

// ------------------------------------------------------------------------------------------------
// Multivariate path for H <= 64 (WaDi QNet / SMD): the input projection of a sample,
// x_n . Wx + b, is the same in every window that contains the sample (W of them) and for both
// candidate states, so it is computed ONCE per sample (xproj_kernel) instead of W*S times, and
// the recurrence (lstm_rec_kernel) starts every step from it.
//
// P layout: [chunk = tile*4 + rg][sample][half][4 floats] where the 4 floats are accumulator
// registers 4rg..4rg+3 of (tile, half) — i.e. the four gates i,j,f,o of one hidden unit.  A wave's
// store (producer) or load (consumer) of one chunk is then 2 KiB contiguous, and the chunk index
// is wave-uniform: the consumer addresses it as a scalar base + one 32-bit lane offset.
struct XprojArgs {
    const float* x;          // [n_samples, F]
    long long n_samples;
    int F;
    const float* wx;         // packed [ceil(I/2)][NT][64] (the flag row, feature F, is skipped here)
    const float* bias;       // packed [NT][2][16]
    float* P;                // [NT*4][n_samples][2][4]
    int nt_total;            // gate tiles of the packed weights (8 or 16); blockIdx.y selects a group of NT
    // optional per-series MinMax (reference env.py:78-79) applied to the samples on the way in
    const long long* samp_off; int n_seg;
    const float* in_scale; const float* in_offset;
};

template <int NT>
__global__ __launch_bounds__(256, 2) void xproj_kernel(XprojArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    const int tile0 = blockIdx.y * NT;                       // H = 128: two passes of 8 gate tiles
    const long long n0 = ((long long)blockIdx.x * 4 + wave) * 32;
    if (n0 >= a.n_samples) return;
    const long long n = n0 + j < a.n_samples ? n0 + j : a.n_samples - 1;
    const float* __restrict__ xr = a.x + n * a.F;
    const bool has_norm = a.in_scale != nullptr;
    float sc = 1.0f, of = 0.0f;
    if (has_norm) {                                          // the series this lane's sample belongs to
        int lo = 0, hi = a.n_seg;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a.samp_off[mid] <= n) lo = mid; else hi = mid; }
        sc = a.in_scale[lo]; of = a.in_offset[lo];
    }
    f32x16 acc[NT];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) acc[tl] = load_bias16(a.bias, tile0 + tl, half);
    const int nks = (a.F + 1) >> 1;
#pragma unroll 2
    for (int ks = 0; ks < nks; ++ks) {
        const int feat = 2 * ks + half;
        float bv = feat < a.F ? xr[feat] : 0.0f;
        if (has_norm && feat < a.F) bv = minmax_apply(bv, sc, of);
        const float* __restrict__ wp = a.wx + ((size_t)ks * a.nt_total + tile0) * 64 + lane;
#pragma unroll
        for (int tl = 0; tl < NT; ++tl) acc[tl] = mfma32(wp[tl * 64], bv, acc[tl]);
    }
    if (n0 + j < a.n_samples) {
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                f32x4 v = {acc[tl][4 * rg], acc[tl][4 * rg + 1], acc[tl][4 * rg + 2], acc[tl][4 * rg + 3]};
                *reinterpret_cast<f32x4*>(a.P + ((((size_t)(tile0 + tl)) * 4 + rg) * a.n_samples + (n0 + j)) * 8 + half * 4) = v;
            }
    }
}

struct RecArgs {
    LstmArgs l;
    const float* P;          // xproj output
    long long n_samples;
    const float* flagw;      // packed [NT][2][16]: the flag row of Wx in accumulator order
    int win_rows;            // projected rows between consecutive windows: 1 (sliding windows of a series) or
                             // W (explicit estimator.predict states: every state brings its own W rows)
    int tile0 = 0;           // lstm_rec_kernel / lstm_rec2_kernel / lstm_h128p_kernel: this launch covers tiles [tile0, l.n_tiles)
};

// 4 waves per workgroup (one per SIMD), each an independent 32-window x 1-candidate recurrence:
// U (<= 64 KiB) is shared through LDS as A fragments, h_t feeds step t+1 straight from registers
// (accumulator order == B-operand order), c never leaves registers, no barriers after the weights
// are staged.
//  * With two candidate states (WaDi rule) the waves of a workgroup are {tile, tile+1} x {a0, a1}:
//    both candidates read the same projections, so the second one hits in the CU's own cache.
//  * A step walks the gate tiles one after the other (one 16-register accumulator, a chain of 4*NT
//    MFMAs, then the cell update of that tile's 8 units), so the projections are needed one tile
//    at a time: a 4-slot ring requests tile tl+4 when tile tl is consumed, half a step ahead.
//  * A fragments come from LDS in groups of 8, one group ahead of their MFMAs, across tile
//    boundaries too, so the MFMA stream does not drain at a boundary.
#ifndef REC_RING
#define REC_RING 4      // projection tiles in flight per wave (<= NT); 2, 4 and 8 measure the same (C3, C4):
                        // the refill is bound by L2/fabric throughput, not by latency
#endif
#ifndef REC_ABL
#define REC_ABL 0     // timing-only ablations (wrong results): 1 no projection refill, 2 no cell update, 3 no flag term, 4 = 1+2,
                      // 5 (rec2) no LDS reads of A fragments, 6 (rec2) the refills touch only gate tiles 0..3: half the L2 footprint
#endif
template <int NT>
__device__ __forceinline__ f32x16 rec_load_p(const char* __restrict__ P, size_t chunk_bytes, unsigned voff, int tl) {
    f32x16 d;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const char* base = P + (size_t)(tl * 4 + rg) * chunk_bytes;          // wave-uniform
        const f32x4 v = *reinterpret_cast<const f32x4*>(base + voff);       // + 32-bit lane offset
        d[4 * rg] = v[0]; d[4 * rg + 1] = v[1]; d[4 * rg + 2] = v[2]; d[4 * rg + 3] = v[3];
    }
    return d;
}

// One step of one wave.  FIRST: h = 0, so only the projections (and the flag term) enter the cell.
template <int NT, bool FLAGGED, bool FIRST>
__device__ __forceinline__ void rec_step(f32x16 (&pq)[REC_RING], float (&aq)[2][8], float (&c)[NT][4], float (&h)[NT][4],
                                         const float* __restrict__ ul, const float* __restrict__ s_fw,
                                         const char* __restrict__ Pb, size_t chunk_bytes, unsigned voff_t,
                                         unsigned voff_n, float fl, int half) {
    float hn[NT][4];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        f32x16 acc = pq[tl % REC_RING];
        // refill the slot: tile tl+4 of this step, or tile tl+4-NT of the next one
#if REC_ABL != 1 && REC_ABL != 4
        pq[tl % REC_RING] = (tl + REC_RING < NT) ? rec_load_p<NT>(Pb, chunk_bytes, voff_t, tl + REC_RING)
                                   : rec_load_p<NT>(Pb, chunk_bytes, voff_n, tl + REC_RING - NT);
#endif
        if (FLAGGED && REC_ABL != 3) {
            const f32x16 wf = load_bias16(s_fw, tl, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = fmaf(fl, wf[r], acc[r]);
        }
        if (!FIRST) {
#pragma unroll
            for (int g = 0; g < NT / 2; ++g) {
                // next group of A fragments: same tile, or group 0 of the next tile (tile 0 after the last)
                const int ng = (g + 1) % (NT / 2), ntl = (g + 1 < NT / 2) ? tl : (tl + 1) % NT;
#pragma unroll
                for (int i = 0; i < 8; ++i) aq[(g + 1) & 1][i] = ul[((ng * 8 + i) * NT + ntl) * 64];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = g * 8 + i;
                    acc = mfma32(aq[g & 1][i], h[kk >> 2][kk & 3], acc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
#if REC_ABL == 2 || REC_ABL == 4
            hn[tl][qd] = (acc[4 * qd] + acc[4 * qd + 1]) + (acc[4 * qd + 2] + acc[4 * qd + 3]);
#else
            hn[tl][qd] = lstm_cell(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3], 0.0f, c[tl][qd]);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) h[tl][qd] = hn[tl][qd];
}

#ifndef REC_WAVES_PER_SIMD
#define REC_WAVES_PER_SIMD 1
#endif
template <int NT, bool FLAGGED>
__global__ __launch_bounds__(256, REC_WAVES_PER_SIMD) void lstm_rec_kernel(RecArgs ra) {
    DRL_CLK
    extern __shared__ __attribute__((aligned(16))) float rec_u[];     // [NT*4][NT][64] | flag row [NT][2][16]
    static_assert(NT % 4 == 0, "the projection ring assumes a multiple of 4 gate tiles");
    const LstmArgs& a = ra.l;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    float* s_fw = rec_u + NT * 4 * NT * 64;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.u);
        f32x4* dst = reinterpret_cast<f32x4*>(rec_u);
        for (int i = threadIdx.x; i < NT * 4 * NT * 16; i += 256) dst[i] = src[i];
        for (int i = threadIdx.x; i < NT * 32; i += 256) s_fw[i] = ra.flagw[i];
    }
    __syncthreads();
    // two candidate states: waves = {tile, tile+1} x {a0, a1} (the tail of a call, and calls too small
    // to fill the chip with one wave per tile: see the host side); otherwise 4 tiles
    const int tile = ra.tile0 + (a.S == 2 ? blockIdx.x * 2 + (wave >> 1) : blockIdx.x * 4 + wave);
    const int cand = a.S == 2 ? (wave & 1) : 0;
    if (tile >= a.n_tiles) return;
    const int s = find_seg(a.seg.tile_off, a.seg.n_seg, tile);
    const long long samp0 = a.seg.samp_off[s];
    const long long Ks = a.seg.win_off[s + 1] - a.seg.win_off[s];
    const long long k0 = (long long)(tile - a.seg.tile_off[s]) * 32;
    const int count = (int)(Ks - k0 < 32 ? Ks - k0 : 32);
    const int jj = j < count ? j : count - 1;
    const long long k = k0 + jj;
    const long long nrow0 = samp0 + a.row0 + k * ra.win_rows;   // projected row of row 0 of this lane's window
    const float* __restrict__ ul = rec_u + lane;
    // one chunk of P is n_samples*32 bytes (< 4 GiB, checked by the host): 32-bit lane offsets
    // against 4*NT uniform chunk bases
    const char* __restrict__ Pb = reinterpret_cast<const char*>(ra.P);
    const size_t chunk_bytes = (size_t)ra.n_samples * 32;
    const unsigned voff0 = (unsigned)((size_t)nrow0 * 32 + half * 16);
    const int W = a.W;

    f32x16 pq[REC_RING];
    float c[NT][4], h[NT][4];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) { c[tl][qd] = 0.0f; h[tl][qd] = 0.0f; }
#pragma unroll
    for (int tl = 0; tl < REC_RING; ++tl) pq[tl] = rec_load_p<NT>(Pb, chunk_bytes, voff0, tl);
    float aq[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) aq[0][i] = ul[(i * NT + 0) * 64];

    auto flag_of = [&](int t) -> float {
        if (!FLAGGED) return 0.0f;
        if (a.flag_mode == 3) return (float)cand;
        return synth_flag(a.flag_mode, k, t, W, cand, a.flags, nrow0 + t);
    };
    // the ring's last refill of the last step re-reads that step (W-1 clamps): harmless, branch-free
    rec_step<NT, FLAGGED, true>(pq, aq, c, h, ul, s_fw, Pb, chunk_bytes, voff0,
                                voff0 + (unsigned)(1 < W ? 1 : 0) * 32u, flag_of(0), half);
    for (int t = 1; t < W; ++t) {
        const int tn = t + 1 < W ? t + 1 : t;
        rec_step<NT, FLAGGED, false>(pq, aq, c, h, ul, s_fw, Pb, chunk_bytes, voff0 + (unsigned)t * 32u,
                                     voff0 + (unsigned)tn * 32u, flag_of(t), half);
    }

    ActEpi epi;
    for (int act = 0; act < a.A; ++act) {
        float part = 0.0f;
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int unit = 8 * tl + 2 * qd + half;
                part = fmaf(h[tl][qd], unit < a.H ? a.w_out[unit * a.A + act] : 0.0f, part);
            }
        part += __shfl_xor(part, 32);
        const float qv = part + a.b_out[act];
        epi.push(act, qv);
        if (half == 0 && j < count)
            a.q[((a.seg.win_off[s] + k0 + j) * a.S + cand) * a.A + act] = qv;
    }
    if (a.S == 1 && half == 0 && j < count) epi.store(a, a.seg.win_off[s] + k0 + j);
}

// Two candidate states per window (DRL_FLAG_BOTH: the flag column is 0 on every row of candidate 0 and
// 1 on every row of candidate 1): one wave runs BOTH recurrences of its 32 windows.  They share the
// projections (loaded once; candidate 1 adds the flag row of Wx) and every A fragment (one LDS read
// feeds two MFMAs), and the two accumulator chains are independent of each other.  Same sums in the
// same order as lstm_rec_kernel<NT, true> with flag 0 / 1 (fmaf(1, w, p) == p + w), bit for bit.
template <int NT, bool FIRST>
__device__ __forceinline__ void rec_step2(f32x16 (&pq)[REC_RING], float (&aq)[2][8], float (&c)[2][NT][4], float (&h)[2][NT][4],
                                          const float* __restrict__ ul, const float* __restrict__ s_fw,
                                          const char* __restrict__ Pb, size_t chunk_bytes, unsigned voff_t,
                                          unsigned voff_n, int half) {
    float hn[2][NT][4];
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        f32x16 acc0 = pq[tl % REC_RING], acc1;
#if REC_ABL == 6
        // half the footprint, the same loads: every refill reads one of gate tiles 0..3 (16 KiB per wave and step)
        pq[tl % REC_RING] = (tl + REC_RING < NT) ? rec_load_p<NT>(Pb, chunk_bytes, voff_t, (tl + REC_RING) & 3)
                                   : rec_load_p<NT>(Pb, chunk_bytes, voff_n, (tl + REC_RING - NT) & 3);
#elif REC_ABL != 1 && REC_ABL != 4
        pq[tl % REC_RING] = (tl + REC_RING < NT) ? rec_load_p<NT>(Pb, chunk_bytes, voff_t, tl + REC_RING)
                                   : rec_load_p<NT>(Pb, chunk_bytes, voff_n, tl + REC_RING - NT);
#endif
        {
            const f32x16 wf = load_bias16(s_fw, tl, half);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = acc0[r] + wf[r];
        }
        if (!FIRST) {
#pragma unroll
            for (int g = 0; g < NT / 2; ++g) {
                const int ng = (g + 1) % (NT / 2), ntl = (g + 1 < NT / 2) ? tl : (tl + 1) % NT;
#if REC_ABL != 5
#pragma unroll
                for (int i = 0; i < 8; ++i) aq[(g + 1) & 1][i] = ul[((ng * 8 + i) * NT + ntl) * 64];
#endif
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kk = g * 8 + i;
                    acc0 = mfma32(aq[g & 1][i], h[0][kk >> 2][kk & 3], acc0);
                    acc1 = mfma32(aq[g & 1][i], h[1][kk >> 2][kk & 3], acc1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
#if REC_ABL == 2 || REC_ABL == 4
            hn[0][tl][qd] = (acc0[4 * qd] + acc0[4 * qd + 1]) + (acc0[4 * qd + 2] + acc0[4 * qd + 3]);
            hn[1][tl][qd] = (acc1[4 * qd] + acc1[4 * qd + 1]) + (acc1[4 * qd + 2] + acc1[4 * qd + 3]);
#else
            hn[0][tl][qd] = lstm_cell(acc0[4 * qd], acc0[4 * qd + 1], acc0[4 * qd + 2], acc0[4 * qd + 3], 0.0f, c[0][tl][qd]);
            hn[1][tl][qd] = lstm_cell(acc1[4 * qd], acc1[4 * qd + 1], acc1[4 * qd + 2], acc1[4 * qd + 3], 0.0f, c[1][tl][qd]);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int cd = 0; cd < 2; ++cd)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) h[cd][tl][qd] = hn[cd][tl][qd];
}

template <int NT>
__global__ __launch_bounds__(256, 1) void lstm_rec2_kernel(RecArgs ra) {
    DRL_CLK
    extern __shared__ __attribute__((aligned(16))) float rec_u[];     // [NT*4][NT][64] | flag row [NT][2][16]
    static_assert(NT % 4 == 0, "the projection ring assumes a multiple of 4 gate tiles");
    const LstmArgs& a = ra.l;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    float* s_fw = rec_u + NT * 4 * NT * 64;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.u);
        f32x4* dst = reinterpret_cast<f32x4*>(rec_u);
        for (int i = threadIdx.x; i < NT * 4 * NT * 16; i += 256) dst[i] = src[i];
        for (int i = threadIdx.x; i < NT * 32; i += 256) s_fw[i] = ra.flagw[i];
    }
    __syncthreads();
    const int tile = ra.tile0 + blockIdx.x * 4 + wave;
    if (tile >= a.n_tiles) return;
    const int s = find_seg(a.seg.tile_off, a.seg.n_seg, tile);
    const long long samp0 = a.seg.samp_off[s];
    const long long Ks = a.seg.win_off[s + 1] - a.seg.win_off[s];
    const long long k0 = (long long)(tile - a.seg.tile_off[s]) * 32;
    const int count = (int)(Ks - k0 < 32 ? Ks - k0 : 32);
    const int jj = j < count ? j : count - 1;
    const long long nrow0 = samp0 + a.row0 + (k0 + jj) * ra.win_rows;
    const float* __restrict__ ul = rec_u + lane;
    const char* __restrict__ Pb = reinterpret_cast<const char*>(ra.P);
    const size_t chunk_bytes = (size_t)ra.n_samples * 32;
    const unsigned voff0 = (unsigned)((size_t)nrow0 * 32 + half * 16);
    const int W = a.W;

    f32x16 pq[REC_RING];
    float c[2][NT][4], h[2][NT][4];
#pragma unroll
    for (int cd = 0; cd < 2; ++cd)
#pragma unroll
        for (int tl = 0; tl < NT; ++tl)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) { c[cd][tl][qd] = 0.0f; h[cd][tl][qd] = 0.0f; }
#pragma unroll
    for (int tl = 0; tl < REC_RING; ++tl) pq[tl] = rec_load_p<NT>(Pb, chunk_bytes, voff0, tl);
    float aq[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) aq[0][i] = ul[(i * NT + 0) * 64];

    rec_step2<NT, true>(pq, aq, c, h, ul, s_fw, Pb, chunk_bytes, voff0, voff0 + (unsigned)(1 < W ? 1 : 0) * 32u, half);
    for (int t = 1; t < W; ++t) {
        const int tn = t + 1 < W ? t + 1 : t;
        rec_step2<NT, false>(pq, aq, c, h, ul, s_fw, Pb, chunk_bytes, voff0 + (unsigned)t * 32u,
                             voff0 + (unsigned)tn * 32u, half);
    }

    // the wave holds the Q of both candidate states: np.argmax over candidate 0's actions (rule 0), or the WaDi
    // rule, Q(s1)[1] > Q(s0)[0] (finalize_kernel's operations)
    ActEpi epi;
    float q1_wadi = 0.0f;
#pragma unroll
    for (int cd = 0; cd < 2; ++cd)
        for (int act = 0; act < a.A; ++act) {
            float part = 0.0f;
#pragma unroll
            for (int tl = 0; tl < NT; ++tl)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int unit = 8 * tl + 2 * qd + half;
                    part = fmaf(h[cd][tl][qd], unit < a.H ? a.w_out[unit * a.A + act] : 0.0f, part);
                }
            part += __shfl_xor(part, 32);
            const float qv = part + a.b_out[act];
            if (cd == 0) epi.push(act, qv);
            else if (act == 1) q1_wadi = qv;
            if (half == 0 && j < count)
                a.q[((a.seg.win_off[s] + k0 + j) * 2 + cd) * a.A + act] = qv;
        }
    if (half == 0 && j < count) {
        if (a.rule == 1) { epi.q1 = q1_wadi; epi.best = q1_wadi > epi.q0 ? 1 : 0; }
        epi.store(a, a.seg.win_off[s] + k0 + j);
    }
}

// Univariate fast path (I = 2, H = 128): one workgroup of 4 waves = 64 consecutive windows in two
// groups of 32.  The recurrent weights U (256 KiB fp32) do not fit the 160 KiB LDS, so they live
// in the CU's register file for the whole kernel: wave w owns gate tiles 4w..4w+3 (32 hidden units
// x 4 gates) = 256 VGPRs/AGPRs of A fragments.  Only h_t (32 windows x 128 units = 16 KiB per
// group) crosses waves, through LDS, already in B-fragment order.  The two groups run half a step
// apart: while the MFMAs of one group run, the gate non-linearities of the other group issue on
// the VALU of the same wave.
#define H128_SV 256
#ifndef H128_UREG
#define H128_UREG 3     // gate tiles (of 4 per wave) whose recurrent weights sit in AGPRs; the rest in LDS
#endif
__device__ __forceinline__ void h128_gemm(f32x16 (&acc)[4], const float (&U)[H128_UREG][64], const float (&wx)[4],
                                          const float* __restrict__ ul, const float* __restrict__ hb,
                                          const float* __restrict__ bias_l, float bx, int lane, int half, int w) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        acc[tl] = load_bias16(bias_l, 4 * w + tl, half);
        acc[tl] = mfma32(wx[tl], bx, acc[tl]);
    }
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) {
        const float b = hb[kk * 64 + lane];
#pragma unroll
        for (int tl = 0; tl < H128_UREG; ++tl) acc[tl] = mfma32(U[tl][kk], b, acc[tl]);
#pragma unroll
        for (int tl = H128_UREG; tl < 4; ++tl) acc[tl] = mfma32(ul[((tl - H128_UREG) * 64 + kk) * 64 + lane], b, acc[tl]);
    }
}

__device__ __forceinline__ void h128_xonly(f32x16 (&acc)[4], const float (&wx)[4],
                                           const float* __restrict__ bias_l, float bx, int half, int w) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        acc[tl] = load_bias16(bias_l, 4 * w + tl, half);
        acc[tl] = mfma32(wx[tl], bx, acc[tl]);
    }
}

#ifndef H128_PW_FENCE
#define H128_PW_FENCE 1
#endif
__device__ __forceinline__ void h128_pointwise(const f32x16 (&acc)[4], float (&c)[4][4], float* __restrict__ hb,
                                               float fb, int lane, int w) {
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float hv = lstm_cell(acc[tl][4 * qd], acc[tl][4 * qd + 1], acc[tl][4 * qd + 2],
                                       acc[tl][4 * qd + 3], fb, c[tl][qd]);
            hb[((4 * w + tl) * 4 + qd) * 64 + lane] = hv;
#if H128_PW_FENCE == 2
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#if H128_PW_FENCE == 1
        __builtin_amdgcn_sched_barrier(0);      // bound the live temporaries: one gate tile at a time
#endif
    }
}

#ifndef H128_ABL
#define H128_ABL 0        // timing-only ablations (wrong results): 2 no barriers, 3 prologue only
#endif
#if H128_ABL == 2
#define H128_SYNC() do { } while (0)
#else
#define H128_SYNC() __syncthreads()
#endif

// NG = window groups of 32 per tile.  2 is the throughput shape (64 windows per tile).  1 halves the
// tile so that a call with no more 32-window tiles than CUs (one Yahoo series, a replay batch) spreads
// over twice as many CUs: same arithmetic per window, half the latency.
template <int NG>
__global__ __launch_bounds__(256, 1) void lstm_h128i2_kernel(LstmArgs a) {
    DRL_CLK
    // separate objects so that alias analysis can tell group A's h from group B's
    __shared__ __attribute__((aligned(16))) float s_hA[64 * 64];   // [k-step][lane]
    __shared__ __attribute__((aligned(16))) float s_hB[64 * 64];
    __shared__ __attribute__((aligned(16))) float s_bias[16 * 2 * 16];
    __shared__ __attribute__((aligned(16))) float s_u[4][(4 - H128_UREG) * 64 * 64 + 4];   // per wave: U of its last tile(s)
    __shared__ float s_sv[H128_SV];          // normalised samples k0 .. k0+64+W
    __shared__ float s_sf[H128_SV];          // explicit flags of those samples
    __shared__ float s_wout[128 * 8];        // Q head, [unit][action]
    __shared__ SegLds s_seg;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    const int cand = blockIdx.y;
    const int W = a.W;
    seg_stage(s_seg, a.seg);
    for (int i = threadIdx.x; i < 128 * a.A; i += 256) s_wout[i] = a.w_out[i];

    // weights to registers, ONCE per workgroup: the grid is persistent (one workgroup per CU walks
    // tiles blockIdx.x, +gridDim.x, ...), because every workgroup of a wave of launches would
    // otherwise pull the same 256 KiB through the same L2 channels at the same moment
    float U[H128_UREG][64], wx[4];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) wx[tl] = a.wx[(4 * w + tl) * 64 + lane];
    // a tile's 64 loads are all in flight before the first value is pinned: load -> wait -> pin, one
    // value at a time, costs 192 round trips to L2 (60 us per workgroup)
#pragma unroll
    for (int tl = 0; tl < H128_UREG; ++tl) {
        float tmp[64];
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) tmp[kk] = a.u[((size_t)kk * 16 + 4 * w + tl) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) U[tl][kk] = pin_agpr(tmp[kk]);
        __builtin_amdgcn_sched_barrier(0);
    }
    float* ul = &s_u[w][0];
    for (int tl = H128_UREG; tl < 4; ++tl) {
#pragma unroll 16
        for (int kk = 0; kk < 64; ++kk)
            ul[((tl - H128_UREG) * 64 + kk) * 64 + lane] = a.u[((size_t)kk * 16 + 4 * w + tl) * 64 + lane];
    }
    for (int i = threadIdx.x; i < 16 * 2 * 16; i += 256) s_bias[i] = a.bias[i];
    const float* bias_l = s_bias;
    float* hA = s_hA;
    float* hB = s_hB;
    const float fb = a.forget_bias;
    __syncthreads();

    for (int tile = a.tile0 + blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
#if H128_ABL == 3
        break;                                  // timing-only: the prologue alone
#endif
        const SegHit sg = seg_lookup(s_seg, a.seg, tile);
        const int s = sg.s;
        const long long samp0 = sg.samp0, nsamp = sg.nsamp, Ks = sg.Ks;
        const long long k0 = (long long)(tile - sg.tile0) * (32 * NG);
        const int count = (int)(Ks - k0 < 32 * NG ? Ks - k0 : 32 * NG);

        // stage the series tile (and explicit flags) into LDS, coalesced, clamped to the series
        // (explicit states, flag_mode < 0, are read row by row instead: see xin)
        if (a.flag_mode >= 0) {
            const bool has_norm = a.seg.in_scale != nullptr;
            const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
            const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
            for (int i = threadIdx.x; i < 32 * NG + W + 1 && i < H128_SV; i += 256) {
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

        // B operand of the input k-step: lanes 0-31 carry the value, lanes 32-63 the flag
        auto xin = [&](int g, int t) -> float {
            if (a.flag_mode < 0) {
                // explicit states [B, W, 2] (estimator.predict batches): row t of this lane's state,
                // requested one step ahead of its use by the loop below
                int jw = g * 32 + j;
                if (jw > count - 1) jw = count - 1;
                return a.x[(samp0 + k0 + jw) * a.win_stride + (long long)t * a.row_stride + half];
            }
            const int li = g * 32 + j + 1 + t;                 // local sample index of row t
            if (half == 0) return s_sv[li];
            if (a.flag_mode == 2) return s_sf[li];
            return synth_flag(a.flag_mode, k0 + g * 32 + j, t, W, cand, nullptr, 0);
        };

        // step 0: h = 0, only the input part
        h128_xonly(accA, wx, bias_l, xin(0, 0), half, w);
        if (NG == 2) h128_xonly(accB, wx, bias_l, xin(1, 0), half, w);
        h128_pointwise(accA, cA, hA, fb, lane, w);
        __syncthreads();
        // the input row of step t+1 is read while step t computes (a global load for explicit states)
        float nxa = xin(0, W > 1 ? 1 : 0), nxb = NG == 2 ? xin(1, W > 1 ? 1 : 0) : 0.0f;
        for (int t = 1; t < W; ++t) {
            const int tn = t + 1 < W ? t + 1 : t;
            if (NG == 2) {
                // phase 1: MFMAs of group A step t  ||  non-linearities of group B step t-1
                const float bxa = nxa;
                nxa = xin(0, tn);
                h128_pointwise(accB, cB, hB, fb, lane, w);
                h128_gemm(accA, U, wx, ul, hA, bias_l, bxa, lane, half, w);
                H128_SYNC();
                // phase 2: MFMAs of group B step t  ||  non-linearities of group A step t
                const float bxb = nxb;
                nxb = xin(1, tn);
                h128_pointwise(accA, cA, hA, fb, lane, w);
                h128_gemm(accB, U, wx, ul, hB, bias_l, bxb, lane, half, w);
                H128_SYNC();
            } else {
                const float bxa = nxa;
                nxa = xin(0, tn);
                h128_gemm(accA, U, wx, ul, hA, bias_l, bxa, lane, half, w);
                H128_SYNC();                    // every wave has read h_{t-1}
                h128_pointwise(accA, cA, hA, fb, lane, w);
                H128_SYNC();                    // h_t is complete
            }
        }
        if (NG == 2) h128_pointwise(accB, cB, hB, fb, lane, w);
        __syncthreads();

        // Q head: waves 0/1 -> groups A/B; lane sums the units of its half, then the halves combine
        if (w < NG) {
            const float* hb = w == 0 ? hA : hB;
            const int jw = w * 32 + j;
            ActEpi epi;
            for (int act = 0; act < a.A; ++act) {
                float part = 0.0f;
#pragma unroll 8
                for (int kk = 0; kk < 64; ++kk)
                    part = fmaf(hb[kk * 64 + lane], s_wout[((kk >> 2) * 8 + 2 * (kk & 3) + half) * a.A + act], part);
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

// Multivariate H = 128 (any input width; e.g. SMD/WaDi sensors with the reference's default
// n_hidden_dim = 128): the structure of lstm_h128i2_kernel — U stationary in the CU's register
// file, h through LDS, two window groups half a step apart, persistent grid — with the input side
// taken from the per-sample projections P of xproj_kernel<16> instead of an input k-step.  A group's
// accumulators are free from its cell update until its next product, so the projections of its
// next step are loaded straight into them in that gap and arrive under the other group's MFMAs.
__device__ __forceinline__ void h128_mm(f32x16 (&acc)[4], const float (&U)[H128_UREG][64], const float* __restrict__ ul,
                                        const float* __restrict__ hb, int lane) {
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) {
        const float b = hb[kk * 64 + lane];
#pragma unroll
        for (int tl = 0; tl < H128_UREG; ++tl) acc[tl] = mfma32(U[tl][kk], b, acc[tl]);
#pragma unroll
        for (int tl = H128_UREG; tl < 4; ++tl) acc[tl] = mfma32(ul[((tl - H128_UREG) * 64 + kk) * 64 + lane], b, acc[tl]);
    }
}

template <int NG>
__global__ __launch_bounds__(256, 1) void lstm_h128p_kernel(RecArgs ra) {
    const LstmArgs& a = ra.l;
    __shared__ __attribute__((aligned(16))) float s_hA[64 * 64];
    __shared__ __attribute__((aligned(16))) float s_hB[64 * 64];
    __shared__ __attribute__((aligned(16))) float s_fw[16 * 2 * 16];   // flag row of Wx, accumulator order
    __shared__ __attribute__((aligned(16))) float s_u[4][(4 - H128_UREG) * 64 * 64 + 4];
    __shared__ float s_wout[128 * 8];
    __shared__ SegLds s_seg;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, j = lane & 31;
    const int cand = blockIdx.y;
    const int W = a.W;
    seg_stage(s_seg, a.seg);
    for (int i = threadIdx.x; i < 128 * a.A; i += 256) s_wout[i] = a.w_out[i];
    for (int i = threadIdx.x; i < 16 * 2 * 16; i += 256) s_fw[i] = ra.flagw[i];

    float U[H128_UREG][64];
#pragma unroll
    for (int tl = 0; tl < H128_UREG; ++tl) {
        float tmp[64];
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) tmp[kk] = a.u[((size_t)kk * 16 + 4 * w + tl) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 64; ++kk) U[tl][kk] = pin_agpr(tmp[kk]);
        __builtin_amdgcn_sched_barrier(0);
    }
    float* ul = &s_u[w][0];
    for (int tl = H128_UREG; tl < 4; ++tl) {
#pragma unroll 16
        for (int kk = 0; kk < 64; ++kk)
            ul[((tl - H128_UREG) * 64 + kk) * 64 + lane] = a.u[((size_t)kk * 16 + 4 * w + tl) * 64 + lane];
    }
    float* hA = s_hA;
    float* hB = s_hB;
    const float fb = a.forget_bias;
    const char* __restrict__ Pb = reinterpret_cast<const char*>(ra.P);
    const size_t chunk_bytes = (size_t)ra.n_samples * 32;
    const bool flagged = a.flag_mode > 0;
    const int wu = __builtin_amdgcn_readfirstlane(w);      // wave-uniform: the chunk bases stay scalar
    __syncthreads();

    for (int tile = ra.tile0 + blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {     // this launch: tiles [tile0, n_tiles)
        const SegHit sg = seg_lookup(s_seg, a.seg, tile);
        const long long k0 = (long long)(tile - sg.tile0) * (32 * NG);
        const int count = (int)(sg.Ks - k0 < 32 * NG ? sg.Ks - k0 : 32 * NG);
        // this lane's window in each group (clamped to the tile's last window) and its row-0 sample
        long long kw[2], nrow[2];
        unsigned voff[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            int jw = g * 32 + j;
            if (jw > count - 1) jw = count - 1;
            kw[g] = k0 + jw;
            nrow[g] = sg.samp0 + a.row0 + kw[g] * ra.win_rows;
            voff[g] = (unsigned)((size_t)nrow[g] * 32 + half * 16);
        }
        // projections of row t of group g into the accumulators of this wave's four gate tiles
        auto load_p = [&](f32x16 (&acc)[4], int g, int t) {
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) acc[tl] = rec_load_p<16>(Pb, chunk_bytes, voff[g] + (unsigned)t * 32u, 4 * wu + tl);
        };
        auto add_flag = [&](f32x16 (&acc)[4], int g, int t) {
            if (!flagged) return;
            const float fl = a.flag_mode == 3 ? (float)cand
                                              : synth_flag(a.flag_mode, kw[g], t, W, cand, a.flags, nrow[g] + t);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                const f32x16 wf = load_bias16(s_fw, 4 * w + tl, half);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tl][r] = fmaf(fl, wf[r], acc[tl][r]);
            }
        };

        f32x16 accA[4], accB[4];
        float cA[4][4], cB[4][4];
#pragma unroll
        for (int tl = 0; tl < 4; ++tl)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) { cA[tl][qd] = 0.0f; cB[tl][qd] = 0.0f; }

        // step 0: h = 0, the projections alone
        load_p(accA, 0, 0);
        if (NG == 2) load_p(accB, 1, 0);
        add_flag(accA, 0, 0);
        h128_pointwise(accA, cA, hA, fb, lane, w);
        if (W > 1) load_p(accA, 0, 1);
        if (NG == 2) add_flag(accB, 1, 0);
        __syncthreads();
        for (int t = 1; t < W; ++t) {
            const int tn = t + 1 < W ? t + 1 : t;
            if (NG == 2) {
                // phase 1: B's cell update of step t-1, B's next projections on their way, A's product
                h128_pointwise(accB, cB, hB, fb, lane, w);
                load_p(accB, 1, t);
                add_flag(accA, 0, t);
                h128_mm(accA, U, ul, hA, lane);
                H128_SYNC();
                // phase 2: the same with the groups swapped
                h128_pointwise(accA, cA, hA, fb, lane, w);
                load_p(accA, 0, tn);
                add_flag(accB, 1, t);
                h128_mm(accB, U, ul, hB, lane);
                H128_SYNC();
            } else {
                add_flag(accA, 0, t);
                h128_mm(accA, U, ul, hA, lane);
                H128_SYNC();
                h128_pointwise(accA, cA, hA, fb, lane, w);
                load_p(accA, 0, tn);
                H128_SYNC();
            }
        }
        if (NG == 2) h128_pointwise(accB, cB, hB, fb, lane, w);
        __syncthreads();

        if (w < NG) {
            const float* hb = w == 0 ? hA : hB;
            const int jw = w * 32 + j;
            ActEpi epi;
            for (int act = 0; act < a.A; ++act) {
                float part = 0.0f;
#pragma unroll 8
                for (int kk = 0; kk < 64; ++kk)
                    part = fmaf(hb[kk * 64 + lane], s_wout[((kk >> 2) * 8 + 2 * (kk & 3) + half) * a.A + act], part);
                part += __shfl_xor(part, 32);
                const float qv = part + a.b_out[act];
                epi.push(act, qv);
                if (half == 0 && jw < count)
                    a.q[((sg.win0 + k0 + jw) * a.S + cand) * a.A + act] = qv;
            }
            if (a.S == 1 && half == 0 && jw < count) epi.store(a, sg.win0 + k0 + jw);
        }
        __syncthreads();
    }
}

// estimator.predict at batch 1..4 — the reference's call at every environment step
// (myasp-A1.py:224-229 policy, env.py:143 candidate states; myasp-wadi.py:326-328 Q of s0 and s1).
// A 32-column MFMA tile would be 97 % empty, so this is a matrix-vector kernel on the VALU: one
// workgroup of 32*NT threads per state, thread r owns gate row r of the packed weights (its H
// recurrent weights in registers), h_t is broadcast from LDS.  The input side, which does not depend
// on h, is done first for all W rows (the state staged in LDS, the input weights streamed once).
// Every sum runs in the order the MFMA kernels use (bias, input features 0..I-1, units 0..H-1), so
// the result is bit-identical to theirs.  W <= GEMV_WMAX.
#define GEMV_WMAX 64
// lane l of a quad gets lane K of its quad (DPP quad_perm: a plain VALU move, no LDS round trip)
template <int K>
__device__ __forceinline__ float quad_bcast(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), K * 0x55, 0xf, 0xf, true));
}
// z += h[16*m + N] * u with h[16*m + (lane & 15)] held in hreg by every row of 16 lanes: the broadcast
// is the DPP operand of the multiply-add itself (row_newbcast: lane N of each row feeds the whole row),
// so a unit costs one VALU instruction and no LDS traffic.  Reading h_t as broadcast ds_read_b128 made
// the LDS return path the bound of a step (8 waves x 32 reads x 1 KiB: 2,048 cycles against 1,024 of
// multiply-adds, measured 2,150).  v_fmac_f32 is the fused multiply-add fmaf is; the chain order is unchanged.
// One asm statement per 16 units: between separate statements the compiler pads every DPP instruction
// with an s_nop (it cannot see that the DPP operand is not the register just written).
__device__ __forceinline__ float fmac_rowbcast16(float z, float hreg, const float* u) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %6 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %10 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %11 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %12 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %13 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %14 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %15 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %16 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %0, %1, %17 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(z)
        : "v"(hreg), "v"(u[0]), "v"(u[1]), "v"(u[2]), "v"(u[3]), "v"(u[4]), "v"(u[5]), "v"(u[6]), "v"(u[7]),
          "v"(u[8]), "v"(u[9]), "v"(u[10]), "v"(u[11]), "v"(u[12]), "v"(u[13]), "v"(u[14]), "v"(u[15]));
    return z;
}
// z += sum over units u = 0 .. HP-1 (ascending) of h[u] * U[u]; hs = h_t in LDS, [HP]
template <int HP>
__device__ __forceinline__ float gemv_chain(float z, const float (&U)[HP], const float* __restrict__ hs, int lane) {
    float hreg[HP / 16];
#pragma unroll
    for (int m = 0; m < HP / 16; ++m) hreg[m] = hs[16 * m + (lane & 15)];
#pragma unroll
    for (int m = 0; m < HP / 16; ++m) z = fmac_rowbcast16(z, hreg[m], &U[16 * m]);
    return z;
}
// lstm_cell for a quad of lanes that holds the four gate pre-activations of one unit (lane g: gate g of
// i, j, f, o): every lane activates its own gate (one v_exp_f32 + one v_rcp_f32 instead of four of each)
// and the quad exchanges the activated values.  Same operations on the same operands as lstm_cell
// (fma(1, a, 0) == a for the sigmoid gates), so the same bits.
__device__ __forceinline__ float lstm_cell_quad(float z, float act_mul, float act_add, float& c) {
    const float v = fmaf(act_mul, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)), act_add);
    const float si = quad_bcast<0>(v), tj = quad_bcast<1>(v), sf = quad_bcast<2>(v), so = quad_bcast<3>(v);
    c = fmaf(sf, c, si * tj);
    return so * fast_tanh(c);
}
template <int NT>
__global__ __launch_bounds__(32 * NT, 1) void lstm_gemv_kernel(LstmArgs a) {
    constexpr int ROWS = 32 * NT, HP = 8 * NT;
    extern __shared__ __attribute__((aligned(16))) float gv_smem[];
    __shared__ __attribute__((aligned(16))) float s_h[2][HP];
    __shared__ float s_part[2 * 8];
    const int r = threadIdx.x;                  // packed gate row: tile r>>5, row-in-tile r&31
    const int tau = r >> 5, i = r & 31;
    const int b = blockIdx.x;                   // the state
    const int W = a.W, I = a.I;
    const int IP = (I + 3) & ~3;                // row stride of the staged state
    float* s_x = gv_smem;                       // [W][IP], zero padded
    float* s_zx = gv_smem + W * IP;             // [W][ROWS]: input part of every step's pre-activations
    {
        const float* __restrict__ x = a.x + (long long)b * a.win_stride;
        // eight elements per thread in flight (the state may sit in mapped host memory: one round trip
        // per batch of loads, not per element)
        const int n = W * IP;
        for (int e0 = r; e0 < n; e0 += 8 * ROWS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + u * ROWS < n ? e0 + u * ROWS : n - 1;
                const int t = e / IP, f = e - t * IP;
                v[u] = x[(long long)t * a.row_stride + (f < I ? f : I - 1)];
                v[u] = f < I ? v[u] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e0 + u * ROWS < n) s_x[e0 + u * ROWS] = v[u];
        }
        if (r < HP) { s_h[0][r] = 0.0f; s_h[1][r] = 0.0f; }
    }
    float U[HP];
#pragma unroll
    for (int u = 0; u < HP; ++u) {
        const int kk = (u >> 3) * 4 + ((u & 7) >> 1), half = u & 1;
        U[u] = a.u[((size_t)kk * NT + tau) * 64 + half * 32 + i];
    }
    __syncthreads();

    // input part: zx[t] = bias + sum_f wx[f] * x[t][f], f ascending — a [rows x I] x [I x W] product, so
    // it runs on the MFMA (columns = the W rows of the state; the chain of k-steps is the same sum in
    // the same order): wave w takes gate tiles w, w + waves, ...; A fragments are the packed wx as it is.
    // (Narrow inputs — the univariate [value, flag] — stay on the VALU: one pass of four features, no
    // MFMA latency, no second barrier; 2 us of a 35 us kernel.)
    if (I <= 4) {
        const float bias = a.bias[(tau * 2 + ((i >> 2) & 1)) * 16 + (i & 3) + 4 * (i >> 3)];
        float w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)             // wx is packed [ceil(I/2)][NT][64]: (f>>1, tau, (f&1)*32 + i); zero beyond I
            w[q] = q < I ? a.wx[((size_t)(q >> 1) * NT + tau) * 64 + (q & 1) * 32 + i] : 0.0f;
        for (int t = 0; t < W; ++t) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(s_x + t * IP);
            float z = bias;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (q < I) z = fmaf(w[q], xv[q], z);
            s_zx[t * ROWS + r] = z;
        }
    } else {
        const int lane = threadIdx.x & 63, half = lane >> 5, j = lane & 31;
        const int nks = (I + 1) >> 1;
        for (int c0 = 0; c0 < W; c0 += 32) {                 // 32 rows of the state per pass (W <= 64: two)
            const int t = c0 + j;
            const float* __restrict__ xrow = s_x + (t < W ? t : W - 1) * IP + half;   // x[t][2*ks + half]
            for (int tl = threadIdx.x >> 6; tl < NT; tl += ROWS / 64) {
                f32x16 acc = load_bias16(a.bias, tl, half);
                for (int ks0 = 0; ks0 < nks; ks0 += 16) {     // 16 fragments requested together
                    float av[16], bv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        av[u] = 0.0f; bv[u] = 0.0f;
                        if (ks0 + u < nks) {                      // wave-uniform
                            av[u] = a.wx[((size_t)(ks0 + u) * NT + tl) * 64 + lane];
                            bv[u] = xrow[2 * (ks0 + u)];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const bool live = ks0 + u < nks;          // wave-uniform
                        const float b = (live && 2 * (ks0 + u) + half < I) ? bv[u] : 0.0f;
                        if (live) acc = mfma32(av[u], b, acc);
                    }
                }
                if (t < W) {
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) s_zx[t * ROWS + tl * 32 + rowmap(rr, half)] = acc[rr];
                }
            }
        }
    }
    __syncthreads();                                          // zx rows are read by other waves' threads

    float c = 0.0f;
    const float act_mul = (i & 3) == 1 ? 2.0f : 1.0f, act_add = (i & 3) == 1 ? -1.0f : 0.0f;   // lane 1 of a quad: tanh
    for (int t = 0; t < W; ++t) {
        const float* __restrict__ hs = s_h[t & 1];
        float z = s_zx[t * ROWS + r];
        if (t > 0) z = gemv_chain<HP>(z, U, hs, r);
        // the four gates of a unit sit in four consecutive lanes
        const float hv = lstm_cell_quad(z, act_mul, act_add, c);
        if ((i & 3) == 0) s_h[(t + 1) & 1][8 * tau + (i >> 2)] = hv;
        __syncthreads();
    }
    // Q head in the order of the MFMA kernels: even and odd units as two chains, then their sum
    const float* __restrict__ hs = s_h[W & 1];
    if (r < 2 * a.A) {
        const int act = r >> 1, half = r & 1;
        float part = 0.0f;
        for (int kk = 0; kk < 4 * NT; ++kk) {
            const int unit = (kk >> 2) * 8 + 2 * (kk & 3) + half;
            part = fmaf(hs[unit], unit < a.H ? a.w_out[unit * a.A + act] : 0.0f, part);
        }
        s_part[r] = part;
    }
    __syncthreads();
    if (r < a.A) a.q[(long long)b * a.A + r] = (s_part[2 * r] + s_part[2 * r + 1]) + a.b_out[r];
}

// ------------------------------------------------------------------------------------------------
// Sequential validator rollout, one persistent workgroup per series (reference myasp-A1.py:471-496
// driving env.reset / env.step, env.py:82-94, 126-151).  The policy's state at cursor
// c = k + W is rows (v[n], flag[n]) of samples n = k+1 .. k+W, where flag[W] = 1 (the initial state's
// last row, myasp-A1.py:126) and flag[n] = the action taken at cursor n - 1 for n > W: the action is
// written straight into the next sample's flag, in LDS.  Every window is a fresh W-step recurrence from
// h = c = 0, so the work per cursor is that of estimator.predict on one state — the matrix-vector form
// of lstm_gemv_kernel (thread r owns gate row r, its recurrent weights in registers, h_t broadcast from
// LDS, every sum in the MFMA kernels' order: bit-identical Q) — but the input side of a sample,
// bias + wx_v*v[n] + wx_f*flag[n], is computed once when the sample enters the window and kept in an LDS
// ring of W rows.  Series run in parallel, one per CU; a series costs N * W dependent steps whatever
// its actions are (the fixed-point iteration of drl_rollout_batch costs one full-parallel pass per
// action change instead: which is faster depends on the data, tools/rollout_time.py).
struct RolloutSeqArgs {
    LstmArgs l;              // weights, W, seg (samp_off / win_off / in_scale / in_offset), x = the series
    int* action;             // [K_total]
    float* q;                // [K_total, A] or null
};
template <int NT>
__global__ __launch_bounds__(32 * NT, 1) void rollout_seq_kernel(RolloutSeqArgs ra) {
    constexpr int ROWS = 32 * NT, HP = 8 * NT;
    const LstmArgs& a = ra.l;
    extern __shared__ __attribute__((aligned(16))) float rs_zx[];   // [W][ROWS]: input part of each sample in the window
    __shared__ __attribute__((aligned(16))) float s_h[2][HP];
    __shared__ float s_part[2 * 8];
    __shared__ int s_act;
    const int r = threadIdx.x;
    const int tau = r >> 5, i = r & 31;
    const int s = blockIdx.x;
    const int W = a.W;
    const long long samp0 = a.seg.samp_off[s];
    const long long nsamp = a.seg.samp_off[s + 1] - samp0;
    const long long K = nsamp - W;
    if (K <= 0) return;                                   // block-uniform
    const long long win0 = a.seg.win_off[s];
    const bool has_norm = a.seg.in_scale != nullptr;
    const float sc = has_norm ? a.seg.in_scale[s] : 1.0f;
    const float of = has_norm ? a.seg.in_offset[s] : 0.0f;
    const float* __restrict__ x = a.x + samp0;
    auto value = [&](long long n) -> float {
        const float v = x[n];
        return has_norm ? minmax_apply(v, sc, of) : v;
    };
    float U[HP];
#pragma unroll
    for (int u = 0; u < HP; ++u) {
        const int kk = (u >> 3) * 4 + ((u & 7) >> 1), half = u & 1;
        U[u] = a.u[((size_t)kk * NT + tau) * 64 + half * 32 + i];
    }
    // packed layouts as in lstm_gemv_kernel: bias [NT][2][16], wx [ceil(I/2)][NT][64] = (f >> 1, tau, (f & 1)*32 + i)
    const float bias = a.bias[(tau * 2 + ((i >> 2) & 1)) * 16 + (i & 3) + 4 * (i >> 3)];
    const float w_v = a.wx[(size_t)tau * 64 + i], w_f = a.wx[(size_t)tau * 64 + 32 + i];
    auto zx_of = [&](float v, float fl) -> float { return fmaf(w_f, fl, fmaf(w_v, v, bias)); };
    for (int n = 1; n <= W; ++n) rs_zx[(n % W) * ROWS + r] = zx_of(value(n), n == W ? 1.0f : 0.0f);
    if (r < HP) { s_h[0][r] = 0.0f; s_h[1][r] = 0.0f; }
    __syncthreads();

    const float act_mul = (i & 3) == 1 ? 2.0f : 1.0f, act_add = (i & 3) == 1 ? -1.0f : 0.0f;   // lane 1 of a quad: tanh
    for (long long k = 0; k < K; ++k) {
        // the sample that enters the window at the next cursor: requested now, used after the action is known
        const float v_next = k + 1 < K ? value(k + W + 1) : 0.0f;
        float c = 0.0f;
        int slot = (int)((k + 1) % W);
        for (int t = 0; t < W; ++t) {
            const float* __restrict__ hs = s_h[t & 1];
            float z = rs_zx[slot * ROWS + r];
            slot = slot + 1 == W ? 0 : slot + 1;
            if (t > 0) z = gemv_chain<HP>(z, U, hs, r);
            const float hv = lstm_cell_quad(z, act_mul, act_add, c);
            if ((i & 3) == 0) s_h[(t + 1) & 1][8 * tau + (i >> 2)] = hv;
            __syncthreads();
        }
        // Q head in the order of the MFMA kernels: even and odd units as two chains, then their sum
        const float* __restrict__ hs = s_h[W & 1];
        if (r < 2 * a.A) {
            const int act = r >> 1, half = r & 1;
            float part = 0.0f;
            for (int kk = 0; kk < 4 * NT; ++kk) {
                const int unit = (kk >> 2) * 8 + 2 * (kk & 3) + half;
                part = fmaf(hs[unit], unit < a.H ? a.w_out[unit * a.A + act] : 0.0f, part);
            }
            s_part[r] = part;
        }
        __syncthreads();
        if (r == 0) {
            // np.argmax, ties -> the first (reference myasp-A1.py:226-227), on the same floats drl_score_batch writes
            int best = 0;
            float bv = 0.0f;
            for (int act = 0; act < a.A; ++act) {
                const float qv = (s_part[2 * act] + s_part[2 * act + 1]) + a.b_out[act];
                if (ra.q) ra.q[(win0 + k) * a.A + act] = qv;
                if (act == 0 || qv > bv) { bv = qv; best = act; }
            }
            ra.action[win0 + k] = best;
            s_act = best;
        }
        __syncthreads();
        // the action becomes the flag of the sample that enters next (env.py:143-147); its slot is the one sample k+1 leaves
        if (k + 1 < K) rs_zx[(int)((k + W + 1) % W) * ROWS + r] = zx_of(v_next, (float)s_act);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// action + AL margin from Q (reference myasp-A1.py:227, 255; myasp-wadi.py:328)
__global__ void finalize_kernel(const float* __restrict__ q, long long K, int S, int A, int rule,
                                int* __restrict__ action, float* __restrict__ margin) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float q0, q1;
    if (rule == 1) { q0 = q[(k * S + 0) * A + 0]; q1 = q[(k * S + 1) * A + 1]; }
    else           { q0 = q[k * S * A + 0];       q1 = q[k * S * A + 1]; }
    if (action) {
        int best = q1 > q0 ? 1 : 0;
        if (rule == 0) {                                  // np.argmax over all A actions
            float bv = best ? q1 : q0;
            for (int i = 2; i < A; ++i) { const float v = q[k * S * A + i]; if (v > bv) { bv = v; best = i; } }
        }
        action[k] = best;
    }
    if (margin) margin[k] = fabsf(q0 - q1);
}

// ------------------------------------------------------------------------------------------------
// Sequential validator rollout (reference myasp-A1.py:481-496 with env.py:126-151) as a
// fixed-point iteration.  The action at cursor c is fed back as the flag of sample c+1, so a_c
// depends on a_{c-W+1..c-1}: a true recurrence over actions.  Scoring all windows with a GUESS
// of the action history and taking the greedy actions gives a' = F(guess); because the
// dependence is causal, the prefix of a' up to the first wrong guess is already final, so
// iterating a <- F(a) extends the correct prefix every pass and stops at a = F(a), the exact
// sequential rollout, after (number of action changes + 1) full-parallel passes.
//
// One thread per window: writes flag[sample of cursor k+W+1] = action[k] and reports a change.
// A changed flag alters the states of the W windows that hold its sample and of no others, so the next pass
// needs only those: when `dirty` is given, their 16-window tiles (seg.tile_off = the 16-window tile table) are
// appended, once each, to `list` (ctl[0] = changed, ctl[1] = tiles in the list).  The windows outside the list
// keep the Q they have, which is what a full pass would compute again.
__global__ void rollout_feedback_kernel(const int* __restrict__ action, float* __restrict__ flags,
                                        SegTab seg, long long K_total, int* __restrict__ ctl,
                                        int* __restrict__ dirty, int* __restrict__ list) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= K_total) return;
    int lo = 0, hi = seg.n_seg;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg.win_off[mid] <= g) lo = mid; else hi = mid; }
    const long long k = g - seg.win_off[lo];
    const long long Ks = seg.win_off[lo + 1] - seg.win_off[lo];
    if (k + 1 >= Ks) return;                               // the last action feeds no later state
    const long long n = seg.samp_off[lo] + seg.W + 1 + k;
    const float nf = (float)action[g];
    if (flags[n] != nf) {
        flags[n] = nf;
        ctl[0] = 1;
        if (dirty) {
            // sample W+1+k of the series is row t of window k' = k+1+(W-1-t): k' = k+1 .. k+W
            const long long k_hi = k + seg.W < Ks - 1 ? k + seg.W : Ks - 1;
            for (long long t16 = (k + 1) >> 4; t16 <= (k_hi >> 4); ++t16) {
                const int tile = seg.tile_off[lo] + (int)t16;
                if (atomicExch(&dirty[tile], 1) == 0) list[atomicAdd(&ctl[1], 1)] = tile;
            }
        }
    }
}
// initial state: flags 0 except the 1 the first state carries in its last row (myasp-A1.py:126)
__global__ void rollout_init_kernel(float* __restrict__ flags, SegTab seg, long long n_total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_total) flags[i] = 0.0f;
}
__global__ void rollout_init1_kernel(float* __restrict__ flags, SegTab seg) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= seg.n_seg) return;
    if (seg.samp_off[s + 1] - seg.samp_off[s] > seg.W) flags[seg.samp_off[s] + seg.W] = 1.0f;
}

// ------------------------------------------------------------------------------------------------
// Active-learning selection (reference myasp-A1.py:250-259): rank windows by margin |q0-q1|,
// skip the already labelled ones, take the N most confused.  Keys are prepared here; the sort is
// a stable device radix sort (ties keep ascending window order).
__global__ void al_prepare_kernel(const float* __restrict__ margin, long long K, float* __restrict__ keys,
                                  int* __restrict__ idx) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float m = margin[k];
    keys[k] = (m == m) ? m : __builtin_huge_valf();      // NaN margins rank last, like np.argsort
    idx[k] = (int)k;
}
__global__ void al_mask_kernel(const int* __restrict__ selected, int n_sel, long long K, float* __restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sel) return;
    const int k = selected[i];
    if (k >= 0 && k < K) keys[k] = __builtin_nanf("");    // sorts after +inf in the radix order of floats
}
