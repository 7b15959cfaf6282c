// libdrlvae: C-ABI (include/drlvae.h) over the CDNA4 kernels in kernels.hip.h.
// Host side: weight re-packing into MFMA fragment order, series segment tables, launches.
#include "../../include/drlvae.h"
#include "kernels.hip.h"
#include "lstm_f16x3.hip.h"
#include "lstm_w16.hip.h"
#include "vae_wide16.hip.h"
#include "vae_wide16pp.hip.h"
#include "vae_wide16s.hip.h"

#include <hipcub/hipcub.hpp>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// ---- scratch of the entry point that allocates nothing (drl_score_batch_dev_noalloc) -------------
// Everything that entry point needs on the device beyond the caller's inputs and outputs lives here: the
// per-series tables, the option tables and the per-sample input projections P of the xproj paths.
// Every GPU the library runs on holds its own copy of the array, at its own address, until the process
// ends.  In host code the NAME is only a handle for hipGetSymbolAddress (drl_create, once per device, with
// that device current); everywhere else the per-device pointer g_scratch_dev[device] is used.
#define DRL_SCRATCH_BYTES (32u << 20)
#define DRL_SCRATCH_SEG_BYTES (64u << 10)        // [0, 64 KiB): per-series tables
#define DRL_SCRATCH_OPT_BYTES (192u << 10)       // [64 KiB, 256 KiB): MinMax / RobustScaler tables
#define DRL_SCRATCH_P_OFFSET (DRL_SCRATCH_SEG_BYTES + DRL_SCRATCH_OPT_BYTES)   // [256 KiB, 32 MiB): projections P
#define DRL_NOALLOC_MAX_SEG 1024
#define DRL_NOALLOC_MAX_D 8192
static __device__ __align__(256) unsigned char scratch[DRL_SCRATCH_BYTES];

namespace {

constexpr int MAX_DEVICES = 64;
unsigned char* g_scratch_dev[MAX_DEVICES] = {};   // device address of `scratch` on each GPU (drl_create)
std::mutex g_scratch_mu;

thread_local std::string g_create_error = "";

// Every site in this library that makes the HIP runtime allocate something — device memory, pinned host
// memory, an event, a stream, an RCCL communicator — bumps this counter (drl_debug_alloc_count): the test
// of the entry point that allocates nothing asserts that it does not move across a call.
std::atomic<long long> g_alloc_count{0};
inline void count_alloc() { g_alloc_count.fetch_add(1, std::memory_order_relaxed); }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct VaeHost {
    bool set = false;
    int n_path = 0;                       // layers on the mean path
    int din[DRL_MAX_VAE_LAYERS], dout[DRL_MAX_VAE_LAYERS], act[DRL_MAX_VAE_LAYERS];
    size_t w_off[DRL_MAX_VAE_LAYERS], b_off[DRL_MAX_VAE_LAYERS];   // float offsets in the pack
    int mu_idx = 0;
    size_t wlv_off = 0, blv_off = 0;
    bool has_lv = false;
    float lv_clip = 0;
    int latent = 0;
    int D = 0;                            // input width
    bool narrow = false;                  // every hidden width <= 64: vae_kernel<2>
    size_t pack_floats = 0;               // size of the chain kernels' weight pack (what vae_kernel<.., true> copies to LDS)
    bool wide16 = false;                  // wide windows (D > 128) with first/last hidden layers of at most 128 units: the vae_wide16* kernels
    size_t w16_off[DRL_MAX_VAE_LAYERS], b16_off[DRL_MAX_VAE_LAYERS];   // its 16-row-tile layout, after the chain pack
    size_t wlv16_off = 0, blv16_off = 0;  // the logvar head in the middle layers' 16-row-tile layout (has_lv && wide16)
};

struct LstmHost {
    bool set = false;
    int I = 0, H = 0, A = 0, NT = 0;
    float forget_bias = 1.0f;
    size_t wx_off = 0, u_off = 0, bias_off = 0, wout_off = 0, bout_off = 0, flagw_off = 0;
    size_t uhi_off = 0, ulo_off = 0;      // f16 hi/lo split of U (H = 128 only), float offsets, 16-B aligned
    bool has_split = false;
    size_t u16_off = 0, wx16_off = 0, bias16_off = 0;   // 16-window-tile layout (I = 2, H > 64)
    bool has16 = false;
};

struct Prof {
    bool on = false;
    struct Pair { hipEvent_t a, b; int which; bool a_shared; };   // a_shared: a is the pair before's b (not returned to the pool twice)
    std::vector<Pair> pending;
    std::vector<hipEvent_t> pool;
    double ms[2] = {0, 0};
    int64_t n[2] = {0, 0};
};

}  // namespace

struct drl_ctx {
    int device = 0;
    int n_cu = 256;               // compute units: size of the persistent grids
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // the VAE kernel and the LSTM kernels of one scoring call are independent: the VAE is launched on a
    // second stream (forked from / joined back into `stream` with events) and fills the CUs the LSTM
    // grid leaves idle in its last round
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // RCCL, loaded on first use (dlopen): the library itself has no link-time dependency on it
    void* rccl_lib = nullptr;
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_size = 0;
    bool force_generic = false;   // DRLVAE_FORCE_GENERIC=1: explicit-state batches always take the fallback kernel (tests)
    int overlap_mode = 1;         // 0 never, 1 large wide-window calls, 2 always (DRLVAE_OVERLAP)
    int gemv_rounds = -1;         // estimator.predict batches of up to this many states per CU take the matrix-vector kernel
                                  // (DRLVAE_GEMV_ROUNDS; -1: 4 for the univariate H > 64 net, 8 for the others)
    int rollout_mode = 0;         // drl_rollout_batch: 0 auto, 1 fixed-point over full-parallel passes, 2 one persistent
                                  // workgroup per series (DRLVAE_ROLLOUT / drl_set_rollout_mode)
    size_t rollout_lds_set = 0;
    const int* tl_dev = nullptr;  // set by drl_rollout_batch around a sparse pass: the 16-window tiles score_impl works on
    int tl_n = 0;
    int rec2_mode = 1;            // two candidate states in one wave: 0 never, 1 whole rounds of tiles, 2 always (DRLVAE_REC2)
    // test and measurement knobs (drl_set_knob; the environment variables DRLVAE_<NAME> are read ONCE, at
    // drl_create — no getenv on any call path)
    long long vw16_pad = 0;       // extra LDS bytes per vae_wide16 workgroup: fewer workgroups per CU
    long long rec_lds_pad = 0;    // the same for lstm_rec / lstm_rec2
    long long pbuf_max = -1;      // >= 0: behave as if a projection buffer above this many bytes could not be had
    long long pipe_min = 0;       // > 0: host inputs of at least this many bytes are pipelined (default 16 MiB)
    long long pipe_chunk = 0;     // > 0: fixed chunk size of the pipelined host path, in samples
    bool no_pipeline = false;     // host path: always the one-shot copy
    bool no_fused_epilogue = false;   // action / margin always by finalize_kernel
    bool rollout_dense = false;   // drl_rollout_batch: full passes only
    bool h128_split = true;       // lstm_h128p: a last round that is at most half full runs as 32-window tiles (tests: off)
    int vae_s = 1;                // wide windows: the software-pipelined kernel (vae_wide16s.hip.h): 1 default: wherever one of its
                                  // workgroups fits a CU's LDS; 0: vae_wide16_kernel (tests, A/B)
    int vae_pp = 0;               // wide windows: the ping-pong kernel (two tiles per workgroup, persistent): 0 never (default:
                                  // measured slower than vae_wide16_kernel), 1 calls of more 64-window
                                  // tiles than CUs, 2 whenever its LDS fits
    bool last_overlap = false;
    std::string err;
    VaeHost vae;
    LstmHost lstm;
    DevBuf vae_w, lstm_w;       // packed weights
    DevBuf opt;                 // per-series / per-column option arrays
    std::vector<float> opt_host;   // what `opt` holds (re-uploaded only when it changes)
    size_t opt_S = 0, opt_P = 0;
    DevBuf pbuf;                // per-sample input projections of the H<=64 LSTM path
    DevBuf al_keys, al_keys2, al_idx, al_idx2, al_tmp, al_sel;   // active-learning selection scratch
    bool rec_attr_set = false;
    int lstm_math = 0;          // 0 = exact fp32 MFMA, 1 = 3-term f16 split for the H=128 univariate kernel
    DevBuf st_in, st_flags, st_eps, st_err, st_q, st_act, st_margin;   // staging of the host entry points
    // the entry point that allocates nothing: this GPU's scratch array, and a fixed pinned staging block
    // (allocated at drl_create) through which its tables reach the array; na_ev marks the last copy out of it
    unsigned char* scratch_dev = nullptr;
    void* na_stage = nullptr;
    hipEvent_t na_ev = nullptr;
    bool na_stage_busy = false;
    // pinned, device-mapped host memory for the inputs and outputs of the per-step calls (vae.predict /
    // estimator.predict on one state): the kernel reads and writes it directly, so such a call is one
    // launch and one stream synchronisation, no copies
    void* pin_io = nullptr;
    size_t pin_io_cap = 0;
    Prof prof;
    const char* last_variant = "none";
    std::string last_kernels;   // the LSTM kernels the last scoring call launched, as rocprofv3 names them, ';'-separated
    int last_tile = 0;   // windows per workgroup tile of the last LSTM launch (0: not a tiled kernel)
    const char* last_vae_variant = "none";
    size_t wide_lds_set = 0, gemv_lds_set = 0;
    bool chain_lds_set = false;
    // The per-series tables of the last few calls stay on the device (a slot is found again by its layout key:
    // the usual case is the same layout scored again).  A slot is rewritten only after the event recorded
    // behind the kernels that read it, so asynchronous calls with changing layouts — the chunks of a
    // pipelined host call — never drain the stream.
    struct SegSlot {
        DevBuf dev;
        void* pinned = nullptr;
        size_t pinned_cap = 0;
        std::vector<int64_t> key;
        long long K = 0;
        int t32 = 0, t64 = 0, t16 = 0;
        hipEvent_t ev = nullptr;
        bool busy = false;      // ev marks the last kernels that read the slot
        bool ever = false;      // the slot has been handed out before (a fresh slot has no reader to wait for)
    };
    static constexpr int SEG_SLOTS = 4;
    SegSlot seg_slots[SEG_SLOTS];
    int seg_next = 0, seg_last = -1;
    // pipelined host entry point: input chunks travel on their own stream, ordered with events
    hipStream_t copy_stream = nullptr;
    hipEvent_t copy_ev[SEG_SLOTS] = {};
};

namespace {

int fail(drl_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), DRL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

int ensure(drl_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes <= b.cap) return DRL_OK;
    // the new block first: a failed grow leaves the old one (and whatever it holds) in place
    size_t cap = std::max<size_t>(bytes, 256);
    void* np = nullptr;
    count_alloc();
    HIP_TRY(c, hipMalloc(&np, cap));
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.cap = cap;
    return DRL_OK;
}

int ensure_pin_io(drl_ctx* c, size_t bytes) {
    if (bytes <= c->pin_io_cap) return DRL_OK;
    if (c->pin_io) { HIP_TRY(c, hipHostFree(c->pin_io)); c->pin_io = nullptr; c->pin_io_cap = 0; }
    size_t cap = std::max<size_t>(bytes, 64 * 1024);
    count_alloc();
    HIP_TRY(c, hipHostMalloc(&c->pin_io, cap, hipHostMallocDefault));
    c->pin_io_cap = cap;
    return DRL_OK;
}
constexpr size_t PIN_IO_MAX = 256 * 1024;   // larger calls are staged with copies (PCIe bandwidth, not latency, matters there)

inline int rowmap_h(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
inline size_t up16(size_t n) { return (n + 15) & ~size_t(15); }
inline size_t up32(size_t n) { return (n + 31) & ~size_t(31); }

// Dense layer K [din,dout] -> A fragments [out_tile][in_tile][16][64]:
// lane l of k-step r holds K[in = it*32 + rowmap(r, l>>5)][out = ot*32 + (l&31)] (0 outside).
// in_major = false: [out_tile][in_tile][16][64] (one OUTPUT tile's weights contiguous: last layer);
// in_major = true:  [in_tile][out_tile][16][64] (one INPUT tile's weights contiguous: first layer).
void pack_dense(const float* K, int din, int dout, float* dst, bool in_major, double scale) {
    const int nin = (din + 31) / 32, nout = (dout + 31) / 32;
    for (int ot = 0; ot < nout; ++ot)
        for (int it = 0; it < nin; ++it)
            for (int r = 0; r < 16; ++r)
                for (int l = 0; l < 64; ++l) {
                    const int in = it * 32 + rowmap_h(r, l >> 5), out = ot * 32 + (l & 31);
                    const size_t blk = in_major ? (size_t)it * nout + ot : (size_t)ot * nin + it;
                    dst[(blk * 16 + r) * 64 + l] =
                        (in < din && out < dout) ? (float)(scale * K[(size_t)in * dout + out]) : 0.0f;
                }
}
// bias [dout] -> [out_tile][2][16]: entry (ot, half, r) = bias[ot*32 + rowmap(r, half)]
void pack_bias(const float* b, int dout, float* dst, double scale) {
    const int nout = (dout + 31) / 32;
    for (int ot = 0; ot < nout; ++ot)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int o = ot * 32 + rowmap_h(r, half);
                dst[(ot * 2 + half) * 16 + r] = o < dout ? (float)(scale * b[o]) : 0.0f;
            }
}

// chain: nothing has been queued on `st` since the end event of the pair before, so that event is this pair's
// start as well (an event record costs the stream ~5 us between two kernels: one per boundary, not two)
void prof_begin(drl_ctx* c, int which, hipStream_t st, bool chain = false) {
    if (!c->prof.on) return;
    Prof::Pair p;
    p.a_shared = chain && !c->prof.pending.empty() && c->prof.pending.back().b;
    for (hipEvent_t* e : {&p.a, &p.b}) {
        if (e == &p.a && p.a_shared) { p.a = c->prof.pending.back().b; continue; }
        if (!c->prof.pool.empty()) { *e = c->prof.pool.back(); c->prof.pool.pop_back(); }
        else { count_alloc(); if (hipEventCreate(e) != hipSuccess) *e = nullptr; }
    }
    p.which = which;
    if (p.a && !p.a_shared) (void)hipEventRecord(p.a, st);
    c->prof.pending.push_back(p);
}
void prof_end(drl_ctx* c, hipStream_t st) {
    if (!c->prof.on || c->prof.pending.empty()) return;
    Prof::Pair& p = c->prof.pending.back();
    if (p.b) (void)hipEventRecord(p.b, st);
}
int prof_collect(drl_ctx* c) {
    for (auto& p : c->prof.pending) {
        if (p.a && p.b) {
            HIP_TRY(c, hipEventSynchronize(p.b));
            float ms = 0;
            HIP_TRY(c, hipEventElapsedTime(&ms, p.a, p.b));
            c->prof.ms[p.which] += ms;
            c->prof.n[p.which] += 1;
        }
        if (p.a && !p.a_shared) c->prof.pool.push_back(p.a);
        if (p.b) c->prof.pool.push_back(p.b);
    }
    c->prof.pending.clear();
    return DRL_OK;
}

struct SegHost {
    long long K_total = 0;
    int tiles32 = 0, tiles64 = 0, tiles16 = 0;
    // device pointers into ctx->seg
    const long long* d_samp = nullptr;
    const long long* d_win = nullptr;
    const int* d_t32 = nullptr;
    const int* d_t64 = nullptr;
    const int* d_t16 = nullptr;
};

// The per-series tables of a call, written to `host` (five arrays of n_seg+1 entries); totals into `out`.
int seg_fill(drl_ctx* c, const int64_t* seg_offsets, int n_seg, int W, bool windows_are_rows, void* host, SegHost& out) {
    const size_t n1 = (size_t)n_seg + 1;
    long long* samp = (long long*)host;
    long long* win = samp + n1;
    int* t32 = (int*)(win + n1);
    int* t64 = t32 + n1;
    int* t16 = t64 + n1;
    long long wacc = 0, a32 = 0, a64 = 0, a16 = 0;
    for (int i = 0; i < n_seg; ++i) {
        const long long N = seg_offsets[i + 1] - seg_offsets[i];
        if (N < 0) return fail(c, DRL_ERR_ARG, "seg_offsets must be non-decreasing (series %d)", i);
        const long long K = windows_are_rows ? N : std::max<long long>(0, N - W);
        samp[i] = seg_offsets[i];
        win[i] = wacc;
        t32[i] = (int)a32;
        t64[i] = (int)a64;
        t16[i] = (int)a16;
        wacc += K;
        a32 += (K + 31) / 32;
        a64 += (K + 63) / 64;
        a16 += (K + 15) / 16;
        if (a16 > 0x7fffffffLL) return fail(c, DRL_ERR_ARG, "too many windows for one call");
    }
    samp[n_seg] = seg_offsets[n_seg];
    win[n_seg] = wacc;
    t32[n_seg] = (int)a32;
    t64[n_seg] = (int)a64;
    t16[n_seg] = (int)a16;
    out.K_total = wacc;
    out.tiles32 = (int)a32;
    out.tiles64 = (int)a64;
    out.tiles16 = (int)a16;
    return DRL_OK;
}
size_t seg_bytes(int n_seg) { return ((size_t)n_seg + 1) * (2 * sizeof(long long) + 3 * sizeof(int)); }
void seg_point(SegHost& out, const void* dev, int n_seg) {
    const size_t n1 = (size_t)n_seg + 1;
    out.d_samp = (const long long*)dev;
    out.d_win = out.d_samp + n1;
    out.d_t32 = (const int*)(out.d_win + n1);
    out.d_t64 = out.d_t32 + n1;
    out.d_t16 = out.d_t64 + n1;
}

// Builds and uploads the per-series tables into one of the context's table slots (found again by layout).
int build_segments(drl_ctx* c, const int64_t* seg_offsets, int n_seg, int W, bool windows_are_rows,
                   SegHost& out) {
    const size_t n1 = (size_t)n_seg + 1;
    const size_t bytes = seg_bytes(n_seg);
    std::vector<int64_t> key(seg_offsets, seg_offsets + n1);
    key.push_back(W);
    key.push_back(windows_are_rows ? 1 : 0);
    for (int i = 0; i < drl_ctx::SEG_SLOTS; ++i) {
        drl_ctx::SegSlot& sl = c->seg_slots[i];
        if (sl.dev.p && sl.key == key) {
            out.K_total = sl.K; out.tiles32 = sl.t32; out.tiles64 = sl.t64; out.tiles16 = sl.t16;
            seg_point(out, sl.dev.p, n_seg);
            c->seg_last = i;
            return DRL_OK;
        }
    }
    drl_ctx::SegSlot& sl = c->seg_slots[c->seg_next];
    c->seg_last = c->seg_next;
    c->seg_next = (c->seg_next + 1) % drl_ctx::SEG_SLOTS;
    sl.key.clear();
    // kernels of an earlier asynchronous call may still read this slot, and its pinned block may still be
    // the source of that call's copy: both are behind the slot's event
    // (a slot never handed out before has neither: the first pipelined call after drl_create overlaps from
    // its first chunk)
    if (sl.busy) { HIP_TRY(c, hipEventSynchronize(sl.ev)); sl.busy = false; }
    else if (sl.ever) HIP_TRY(c, hipStreamSynchronize(c->stream));      // a user that recorded nothing (synchronous entry points: a no-op)
    sl.ever = true;
    if (bytes > sl.pinned_cap) {
        if (sl.pinned) { HIP_TRY(c, hipHostFree(sl.pinned)); sl.pinned = nullptr; sl.pinned_cap = 0; }
        const size_t cap = std::max<size_t>(bytes * 2, 4096);
        count_alloc();
        HIP_TRY(c, hipHostMalloc(&sl.pinned, cap, hipHostMallocDefault));
        sl.pinned_cap = cap;
    }
    int rc = ensure(c, sl.dev, bytes);
    if (rc) return rc;
    if ((rc = seg_fill(c, seg_offsets, n_seg, W, windows_are_rows, sl.pinned, out))) return rc;
    HIP_TRY(c, hipMemcpyAsync(sl.dev.p, sl.pinned, bytes, hipMemcpyHostToDevice, c->stream));
    seg_point(out, sl.dev.p, n_seg);
    sl.key = key; sl.K = out.K_total; sl.t32 = out.tiles32; sl.t64 = out.tiles64; sl.t16 = out.tiles16;
    return DRL_OK;
}
// Marks the slot the current call used as busy until everything queued on the stream so far has run.
int seg_mark(drl_ctx* c) {
    if (c->seg_last < 0) return DRL_OK;
    drl_ctx::SegSlot& sl = c->seg_slots[c->seg_last];
    if (!sl.ev) return DRL_OK;
    HIP_TRY(c, hipEventRecord(sl.ev, c->stream));
    sl.busy = true;
    return DRL_OK;
}

void fill_vae_args(drl_ctx* c, VaeArgs& a) {
    const VaeHost& v = c->vae;
    const float* base = (const float*)c->vae_w.p;
    a.n = v.n_path;
    for (int p = 0; p < v.n_path; ++p) {
        a.din[p] = v.din[p]; a.dout[p] = v.dout[p]; a.act[p] = v.act[p];
        a.w[p] = base + v.w_off[p];
        a.b[p] = base + v.b_off[p];
    }
    a.mu_idx = v.mu_idx;
    a.w_lv = v.has_lv ? base + v.wlv_off : nullptr;
    a.b_lv = v.has_lv ? base + v.blv_off : nullptr;
    a.lv_clip = v.lv_clip;
    a.pack = base;
    a.pack_floats = (int)v.pack_floats;
}

// The chain kernel: with its weights copied to LDS first when the call is at most one workgroup per
// CU (nothing else would hide the weight-load latency) and the pack fits.
#ifndef DRL_CHAIN_LDS_MAX
#define DRL_CHAIN_LDS_MAX (128 * 1024)     // 0 (build flag): never, for A/B timing
#endif
constexpr size_t CHAIN_LDS_MAX = DRL_CHAIN_LDS_MAX;
int launch_vae_chain(drl_ctx* c, const VaeArgs& a, int tiles32, hipStream_t st) {
    const dim3 grid((tiles32 + 3) / 4);
    const size_t pack_bytes = c->vae.pack_floats * sizeof(float);
    if ((int)grid.x <= c->n_cu && pack_bytes <= CHAIN_LDS_MAX) {
        if (!c->chain_lds_set) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_kernel<2, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_MAX));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_kernel<DRL_VAE_MAX_TILES, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)CHAIN_LDS_MAX));
            c->chain_lds_set = true;
        }
        if (c->vae.narrow) hipLaunchKernelGGL((vae_kernel<2, true>), grid, dim3(256), pack_bytes, st, a);
        else hipLaunchKernelGGL((vae_kernel<DRL_VAE_MAX_TILES, true>), grid, dim3(256), pack_bytes, st, a);
        c->last_vae_variant = "chain/lds";
    } else {
        if (c->vae.narrow) hipLaunchKernelGGL((vae_kernel<2, false>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((vae_kernel<DRL_VAE_MAX_TILES, false>), grid, dim3(256), 0, st, a);
        c->last_vae_variant = "chain";
    }
    return DRL_OK;
}

void fill_lstm_args(drl_ctx* c, LstmArgs& a) {
    const LstmHost& l = c->lstm;
    const float* base = (const float*)c->lstm_w.p;
    a.I = l.I; a.H = l.H; a.A = l.A;
    a.forget_bias = l.forget_bias;
    a.wx = base + l.wx_off;
    a.u = base + l.u_off;
    a.bias = base + l.bias_off;
    a.w_out = base + l.wout_off;
    a.b_out = base + l.bout_off;
}

}  // namespace

// =================================================================================================
extern "C" {

// DRL_SRC_HASH: sha256 (first 16 hex digits) over the library's sources, passed in by csrc/Makefile; bench.py
// and tests/test_abi.py compare it with the hash of the tree they run from, so a stale prebuilt library shows.
#ifndef DRL_SRC_HASH
#define DRL_SRC_HASH "unknown"
#endif
#define DRL_STR2(x) #x
#define DRL_STR(x) DRL_STR2(x)
const char* drl_version(void) { return "drlvae 0.3 (gfx950; abi " DRL_STR(DRL_ABI_REVISION) "; src " DRL_SRC_HASH ")"; }

int drl_create(int device_id, drl_ctx** out) {
    if (!out) return fail(nullptr, DRL_ERR_ARG, "drl_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, DRL_ERR_HIP, "drl_create: no HIP device (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, DRL_ERR_ARG, "drl_create: device %d out of range (0..%d)", device_id, n - 1);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail(nullptr, DRL_ERR_HIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, DRL_ERR_HIP, "drl_create: device %d is %s; the kernels are built for gfx950 only",
                    device_id, prop.gcnArchName);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return fail(nullptr, DRL_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(e));
    drl_ctx* c = new drl_ctx();
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    count_alloc();      // set-up: streams, events, the pinned staging block below
    e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(nullptr, DRL_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    c->stream = c->own_stream;
    // the second stream carries the VAE kernel beside the LSTM kernels; DRLVAE_AUX_PRIO=low|high creates it
    // at the lowest / highest stream priority (measurement knob; default: the same priority as the main stream)
    hipError_t aux_e;
    {
        const char* pe = getenv("DRLVAE_AUX_PRIO");
        int least = 0, greatest = 0;
        if (pe && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
            aux_e = hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, pe[0] == 'h' ? greatest : least);
        else
            aux_e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    }
    if (aux_e != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        c->overlap_mode = 0;                    // degrade to one stream, never fail creation for it
        c->aux_stream = nullptr;
    }
    for (int i = 0; i < drl_ctx::SEG_SLOTS; ++i) {
        if (hipEventCreateWithFlags(&c->seg_slots[i].ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->seg_slots[i].ev = nullptr; }
        if (hipEventCreateWithFlags(&c->copy_ev[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->copy_ev[i] = nullptr; }
    }
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); c->copy_stream = nullptr; }
    // this GPU's copy of the scratch array (once per device, with the device current: hipSetDevice above),
    // and the fixed pinned block its tables travel through — set-up allocations; the entry point makes none
    {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (device_id < MAX_DEVICES && !g_scratch_dev[device_id]) {
            void* p = nullptr;
            e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
            if (e != hipSuccess || !p) {
                drl_destroy(c);
                return fail(nullptr, DRL_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", device_id, hipGetErrorString(e));
            }
            g_scratch_dev[device_id] = (unsigned char*)p;
        }
        c->scratch_dev = device_id < MAX_DEVICES ? g_scratch_dev[device_id] : nullptr;
    }
    if (hipHostMalloc(&c->na_stage, DRL_SCRATCH_P_OFFSET, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->na_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (c->na_stage) { (void)hipHostFree(c->na_stage); c->na_stage = nullptr; }
        c->scratch_dev = nullptr;               // drl_score_batch_dev_noalloc reports DRL_ERR_STATE
    }
    // Dynamic-LDS limits of every kernel that asks for more than 64 KiB, once per context (= per device) here, so
    // that no scoring call — the one that allocates nothing included — changes a function attribute on its first
    // use.  Should one of them fail, the launch sites still set theirs lazily and report the error there.
    {
        auto attr = [](const void* f, size_t bytes) {
            return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
        };
        const bool ok =
            attr(reinterpret_cast<const void*>(vae_kernel<2, true>), CHAIN_LDS_MAX) &&
            attr(reinterpret_cast<const void*>(vae_kernel<DRL_VAE_MAX_TILES, true>), CHAIN_LDS_MAX) &&
            attr(reinterpret_cast<const void*>(vae_wide16_kernel<true>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16_kernel<false>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16s_kernel<true, 2, 8>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16s_kernel<false, 2, 8>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16s_kernel<true, 2, 4>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16s_kernel<false, 2, 4>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16pp_kernel<true>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(vae_wide16pp_kernel<false>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(lstm_rec_kernel<8, true>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(lstm_rec_kernel<8, false>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(lstm_rec2_kernel<8>), 160 * 1024) &&
            attr(reinterpret_cast<const void*>(lstm_gemv_kernel<8>), 150 * 1024) &&
            attr(reinterpret_cast<const void*>(lstm_gemv_kernel<16>), 150 * 1024) &&
            attr(reinterpret_cast<const void*>(rollout_seq_kernel<8>), 150 * 1024) &&
            attr(reinterpret_cast<const void*>(rollout_seq_kernel<16>), 150 * 1024);
        if (ok) {
            c->chain_lds_set = true; c->rec_attr_set = true;
            c->wide_lds_set = 160 * 1024; c->gemv_lds_set = 150 * 1024; c->rollout_lds_set = 150 * 1024;
        } else (void)hipGetLastError();
    }
    // the environment is read here and nowhere else
    for (const char* name : {"force_generic", "overlap", "rec2", "rollout", "gemv_rounds", "vw16_pad", "rec_lds_pad", "pbuf_max",
                             "pipe_min", "pipe_chunk", "no_pipeline", "no_fused_epilogue", "rollout_dense", "vae_pp", "vae_s", "h128_split"}) {
        std::string var = "DRLVAE_";
        for (const char* p = name; *p; ++p) var += (char)toupper((unsigned char)*p);
        if (const char* e = getenv(var.c_str())) (void)drl_set_knob(c, name, atoll(e));
    }
    *out = c;
    return DRL_OK;
}

void drl_destroy(drl_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)drl_comm_destroy(c);
    for (auto& p : c->prof.pending) { if (p.a && !p.a_shared) (void)hipEventDestroy(p.a); if (p.b) (void)hipEventDestroy(p.b); }
    for (auto e : c->prof.pool) (void)hipEventDestroy(e);
    for (auto& sl : c->seg_slots) {
        if (sl.dev.p) (void)hipFree(sl.dev.p);
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    for (auto e : c->copy_ev) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    for (DevBuf* b : {&c->vae_w, &c->lstm_w, &c->opt, &c->pbuf, &c->al_keys, &c->al_keys2, &c->al_idx, &c->al_idx2, &c->al_tmp, &c->al_sel, &c->st_in, &c->st_flags, &c->st_eps,
                      &c->st_err, &c->st_q, &c->st_act, &c->st_margin})
        if (b->p) (void)hipFree(b->p);
    if (c->pin_io) (void)hipHostFree(c->pin_io);
    if (c->na_stage) (void)hipHostFree(c->na_stage);
    if (c->na_ev) (void)hipEventDestroy(c->na_ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* drl_last_error(drl_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int drl_set_stream(drl_ctx* c, void* s) {
    if (!c) return DRL_ERR_ARG;
    hipStream_t ns = s ? (hipStream_t)s : c->own_stream;
    if (ns != c->stream) {
        // work queued on the old stream still uses the context's buffers: the two streams are not ordered
        HIP_TRY(c, hipSetDevice(c->device));
        if (hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();   // a stream the caller already destroyed
        c->stream = ns;
    }
    return DRL_OK;
}

int drl_set_lstm_math(drl_ctx* c, int mode) {
    if (!c) return DRL_ERR_ARG;
    if (mode != 0 && mode != 1) return fail(c, DRL_ERR_ARG, "drl_set_lstm_math: mode %d (0 = fp32, 1 = f16x3)", mode);
    c->lstm_math = mode;
    return DRL_OK;
}

int drl_set_rollout_mode(drl_ctx* c, int mode) {
    if (!c) return DRL_ERR_ARG;
    if (mode < 0 || mode > 2) return fail(c, DRL_ERR_ARG, "drl_set_rollout_mode: mode %d (0 auto, 1 fixed point, 2 sequential kernel)", mode);
    c->rollout_mode = mode;
    return DRL_OK;
}

int drl_set_knob(drl_ctx* c, const char* name, long long v) {
    if (!c || !name) return DRL_ERR_ARG;
    const std::string n = name;
    if (n == "force_generic") c->force_generic = v != 0;
    else if (n == "overlap") { if (c->aux_stream) c->overlap_mode = (int)std::min<long long>(2, std::max<long long>(0, v)); }
    else if (n == "rec2") c->rec2_mode = (int)std::min<long long>(2, std::max<long long>(0, v));
    else if (n == "rollout") c->rollout_mode = (int)std::min<long long>(2, std::max<long long>(0, v));
    else if (n == "gemv_rounds") c->gemv_rounds = (int)std::max<long long>(-1, v);
    else if (n == "vw16_pad") c->vw16_pad = std::max<long long>(0, v);
    else if (n == "rec_lds_pad") c->rec_lds_pad = std::max<long long>(0, v);
    else if (n == "pbuf_max") c->pbuf_max = v;
    else if (n == "pipe_min") c->pipe_min = std::max<long long>(0, v);
    else if (n == "pipe_chunk") c->pipe_chunk = std::max<long long>(0, v);
    else if (n == "no_pipeline") c->no_pipeline = v != 0;
    else if (n == "no_fused_epilogue") c->no_fused_epilogue = v != 0;
    else if (n == "rollout_dense") c->rollout_dense = v != 0;
    else if (n == "vae_s") c->vae_s = (int)v;
    else if (n == "h128_split") c->h128_split = v != 0;
    else if (n == "vae_pp") c->vae_pp = (int)std::min<long long>(2, std::max<long long>(0, v));
    else return fail(c, DRL_ERR_ARG, "drl_set_knob: unknown knob '%s'", name);
    return DRL_OK;
}

#ifdef VW16_STAMPS
int drl_debug_stamps16(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vw16_dbg), std::min<size_t>((size_t)n, 64 * 5) * sizeof(unsigned long long));
}
#endif
#ifdef DRL_CLK_STAMPS
int drl_debug_clock(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(drl_clk_dbg), 4 * sizeof(unsigned long long));
}
#endif
#ifdef VWS_STAMPS
int drl_debug_stamps_s(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vws_dbg), std::min<size_t>((size_t)n, 64 * 4) * sizeof(unsigned long long));
}
#endif
#ifdef VWP_STAMPS
int drl_debug_stamps(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(vwp_dbg), std::min<size_t>((size_t)n, 2 * 64 * 3) * sizeof(unsigned long long));
}
#endif

long long drl_debug_alloc_count(void) { return g_alloc_count.load(std::memory_order_relaxed); }

int drl_synchronize(drl_ctx* c) {
    if (!c) return DRL_ERR_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DRL_OK;
}

int drl_set_vae(drl_ctx* c, int n_layers, const int32_t* in_dims, const int32_t* out_dims,
                const int32_t* acts, int mu_layer, int logvar_layer,
                const float* const* kernels, const float* const* biases, float logvar_clip) {
    if (!c) return DRL_ERR_ARG;
    if (!in_dims || !out_dims || !acts || !kernels || !biases)
        return fail(c, DRL_ERR_ARG, "drl_set_vae: null argument");
    if (n_layers < 2 || n_layers > DRL_MAX_VAE_LAYERS + 1)
        return fail(c, DRL_ERR_ARG, "drl_set_vae: n_layers %d out of range", n_layers);
    if (mu_layer < 0 || mu_layer >= n_layers - 1)
        return fail(c, DRL_ERR_ARG, "drl_set_vae: mu_layer %d out of range", mu_layer);
    if (logvar_layer >= n_layers || logvar_layer == mu_layer)
        return fail(c, DRL_ERR_ARG, "drl_set_vae: bad logvar_layer %d", logvar_layer);
    for (int l = 0; l < n_layers; ++l) {
        if (in_dims[l] < 1 || out_dims[l] < 1 || !kernels[l] || !biases[l])
            return fail(c, DRL_ERR_ARG, "drl_set_vae: layer %d has bad dims or null weights", l);
        if (acts[l] < 0 || acts[l] > 2) return fail(c, DRL_ERR_ARG, "drl_set_vae: layer %d activation %d", l, acts[l]);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    VaeHost v;
    std::vector<int> path;
    for (int l = 0; l < n_layers; ++l) if (l != logvar_layer) path.push_back(l);
    v.n_path = (int)path.size();
    if (v.n_path < 2 || v.n_path > DRL_MAX_VAE_LAYERS)
        return fail(c, DRL_ERR_UNSUPPORTED, "drl_set_vae: %d layers on the mean path (2..%d supported)", v.n_path,
                    DRL_MAX_VAE_LAYERS);
    size_t off = 0;
    for (int p = 0; p < v.n_path; ++p) {
        const int l = path[p];
        v.din[p] = in_dims[l]; v.dout[p] = out_dims[l]; v.act[p] = acts[l];
        if (l == mu_layer) v.mu_idx = p;
        if (p > 0 && v.din[p] != v.dout[p - 1])
            return fail(c, DRL_ERR_ARG, "drl_set_vae: layer %d input %d != previous output %d", l, v.din[p], v.dout[p - 1]);
        if (p < v.n_path - 1 && v.dout[p] > 32 * DRL_VAE_MAX_TILES)
            return fail(c, DRL_ERR_UNSUPPORTED, "drl_set_vae: hidden width %d > %d", v.dout[p], 32 * DRL_VAE_MAX_TILES);
        const size_t nin = (v.din[p] + 31) / 32, nout = (v.dout[p] + 31) / 32;
        v.w_off[p] = off; off += nin * nout * 1024;
        v.b_off[p] = off; off += up16(nout * 32);
    }
    v.D = v.din[0];
    v.narrow = true;
    for (int p = 0; p + 1 < v.n_path; ++p) v.narrow = v.narrow && v.dout[p] <= 64;
    if (v.dout[v.n_path - 1] != v.D)
        return fail(c, DRL_ERR_ARG, "drl_set_vae: output width %d != input width %d", v.dout[v.n_path - 1], v.D);
    v.latent = out_dims[mu_layer];
    v.has_lv = logvar_layer >= 0;
    if (v.has_lv) {
        if (in_dims[logvar_layer] != in_dims[mu_layer] || out_dims[logvar_layer] != v.latent)
            return fail(c, DRL_ERR_ARG, "drl_set_vae: logvar layer shape differs from mu layer");
        const size_t nin = (in_dims[logvar_layer] + 31) / 32, nout = (v.latent + 31) / 32;
        v.wlv_off = off; off += nin * nout * 1024;
        v.blv_off = off; off += up16(nout * 32);
    }
    v.lv_clip = std::isfinite(logvar_clip) ? logvar_clip : 0.0f;
    v.pack_floats = off;
    // wide windows (vae_wide16_kernel): a second copy of the path's weights in 16-row-tile order
    const int last = v.n_path - 1;
    // (any hidden width up to 128: the 16-row-tile layout is zero-padded to 8 tiles.  Until round 3 widths up to 96 took the
    // chain kernel, which measures 4.2 ms at 64 and 5.2 ms at 96 hidden units where this path takes 2.4 ms at 128.)
    v.wide16 = v.D > 128 && v.dout[0] <= 128 && v.din[last] <= 128;
    for (int p = 1; p < last && v.wide16; ++p) v.wide16 = v.din[p] <= 128 && v.dout[p] <= 128;
    if (v.wide16) {
        const size_t nslab = (v.D + 31) / 32;
        for (int p = 0; p < v.n_path; ++p) {
            const size_t nin16 = (v.din[p] + 15) / 16, nout16 = (v.dout[p] + 15) / 16;
            v.w16_off[p] = off;
            off += (p == 0 || p == last) ? nslab * VW16_SLAB : nout16 * nin16 * 4 * 64;
            v.b16_off[p] = off;
            off += (p == last) ? nslab * 32 : up16(nout16 * 16);
        }
        if (v.has_lv) {                      // after the path's layers: the middle layers stay contiguous
            const size_t nin16 = (in_dims[logvar_layer] + 15) / 16, nout16 = (v.latent + 15) / 16;
            v.wlv16_off = off; off += nout16 * nin16 * 4 * 64;
            v.blv16_off = off; off += up16(nout16 * 16);
        }
    }
    std::vector<float> pack(off, 0.0f);
    for (int p = 0; p < v.n_path; ++p) {
        const int l = path[p];
        // the output sigmoid becomes 1/(1+2^y): fold -log2(e) into the last layer (act code 3)
        double scale = 1.0;
        if (p == v.n_path - 1 && v.act[p] == DRL_ACT_SIGMOID) { scale = -1.4426950408889634; v.act[p] = 3; }
        pack_dense(kernels[l], v.din[p], v.dout[p], pack.data() + v.w_off[p], p == 0, scale);
        pack_bias(biases[l], v.dout[p], pack.data() + v.b_off[p], scale);
    }
    if (v.has_lv) {
        pack_dense(kernels[logvar_layer], in_dims[logvar_layer], v.latent, pack.data() + v.wlv_off, false, 1.0);
        pack_bias(biases[logvar_layer], v.latent, pack.data() + v.blv_off, 1.0);
    }
    if (v.wide16) {
        // layouts: vae_wide16.hip.h (lane l: q = l >> 4 is the k index, l & 15 the output row of a 16-row tile)
        const int D = v.D, nslab = (D + 31) / 32;
        for (int p = 0; p < v.n_path; ++p) {
            const float* K = kernels[path[p]];
            const float* B = biases[path[p]];
            const int din = v.din[p], dout = v.dout[p];
            // v.act[last] was rewritten to 3 above when the output sigmoid is folded: same scale here
            const double scale = (p == last && v.act[p] == 3) ? -1.4426950408889634 : 1.0;
            float* w = pack.data() + v.w16_off[p];
            float* b = pack.data() + v.b16_off[p];
            auto at = [&](int in, int out) -> float {
                return (in < din && out < dout) ? (float)(scale * K[(size_t)in * dout + out]) : 0.0f;
            };
            if (p == 0) {
                for (int it = 0; it < nslab; ++it)
                    for (int T = 0; T < 8; ++T)
                        for (int kg = 0; kg < 2; ++kg)
                            for (int l = 0; l < 64; ++l)
                                for (int j = 0; j < 4; ++j)
                                    w[(size_t)it * VW16_SLAB + (((T * 2 + kg) * 64 + l) * 4 + j)] =
                                        at(it * 32 + 4 * (4 * kg + j) + (l >> 4), 16 * T + (l & 15));
            } else if (p == last) {
                for (int ot = 0; ot < nslab; ++ot)
                    for (int T2 = 0; T2 < 2; ++T2)
                        for (int Tin = 0; Tin < 8; ++Tin)
                            for (int l = 0; l < 64; ++l)
                                for (int r = 0; r < 4; ++r)
                                    w[(size_t)ot * VW16_SLAB + (((T2 * 8 + Tin) * 64 + l) * 4 + r)] =
                                        at(16 * Tin + 4 * (l >> 4) + r, ot * 32 + 16 * T2 + (l & 15));
                for (int d = 0; d < nslab * 32; ++d) b[d] = d < dout ? (float)(scale * B[d]) : 0.0f;
            } else {
                const int nin16 = (din + 15) / 16, nout16 = (dout + 15) / 16;
                for (int ot = 0; ot < nout16; ++ot)
                    for (int Tin = 0; Tin < nin16; ++Tin)
                        for (int r = 0; r < 4; ++r)
                            for (int l = 0; l < 64; ++l)
                                w[(((size_t)ot * nin16 + Tin) * 4 + r) * 64 + l] = at(16 * Tin + 4 * (l >> 4) + r, 16 * ot + (l & 15));
            }
            if (p != last) {
                const int nout16 = (dout + 15) / 16;
                for (int ot = 0; ot < nout16; ++ot)
                    for (int qq = 0; qq < 4; ++qq)
                        for (int r = 0; r < 4; ++r) {
                            const int o = 16 * ot + 4 * qq + r;
                            b[(ot * 4 + qq) * 4 + r] = o < dout ? (float)(scale * B[o]) : 0.0f;
                        }
            }
        }
        if (v.has_lv) {                      // the logvar head, packed like a middle layer
            const float* K = kernels[logvar_layer];
            const float* B = biases[logvar_layer];
            const int din = in_dims[logvar_layer], dout = v.latent;
            const int nin16 = (din + 15) / 16, nout16 = (dout + 15) / 16;
            float* w = pack.data() + v.wlv16_off;
            float* b = pack.data() + v.blv16_off;
            for (int ot = 0; ot < nout16; ++ot)
                for (int Tin = 0; Tin < nin16; ++Tin)
                    for (int r = 0; r < 4; ++r)
                        for (int l = 0; l < 64; ++l) {
                            const int in = 16 * Tin + 4 * (l >> 4) + r, out = 16 * ot + (l & 15);
                            w[(((size_t)ot * nin16 + Tin) * 4 + r) * 64 + l] = (in < din && out < dout) ? K[(size_t)in * dout + out] : 0.0f;
                        }
            for (int ot = 0; ot < nout16; ++ot)
                for (int qq = 0; qq < 4; ++qq)
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * ot + 4 * qq + r;
                        b[(ot * 4 + qq) * 4 + r] = o < dout ? B[o] : 0.0f;
                    }
        }
    }
    int rc = ensure(c, c->vae_w, off * sizeof(float));
    if (rc) { c->vae.set = false; return rc; }      // the old pack may be gone: no stale weights
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->vae_w.p, pack.data(), off * sizeof(float), hipMemcpyHostToDevice));
    v.set = true;
    c->vae = v;
    return DRL_OK;
}

int drl_set_lstm(drl_ctx* c, int in_dim, int hidden, int n_actions, float forget_bias,
                 const float* kernel, const float* bias, const float* w_out, const float* b_out) {
    if (!c) return DRL_ERR_ARG;
    if (!kernel || !bias || !w_out || !b_out) return fail(c, DRL_ERR_ARG, "drl_set_lstm: null argument");
    if (in_dim < 1 || n_actions < 2 || n_actions > 8)
        return fail(c, DRL_ERR_ARG, "drl_set_lstm: in_dim %d / n_actions %d out of range", in_dim, n_actions);
    // any width up to 128: the packs are zero-padded to 64 or 128 units, and a padded unit stays at c = h = 0
    // (all four pre-activations 0: sigmoid(0) * tanh(0) = 0), so it adds exact zeros to every sum
    if (hidden < 1 || hidden > 128)
        return fail(c, DRL_ERR_UNSUPPORTED, "drl_set_lstm: hidden %d (1..128 supported)", hidden);
    HIP_TRY(c, hipSetDevice(c->device));
    LstmHost l;
    l.I = in_dim; l.H = hidden; l.A = n_actions; l.forget_bias = forget_bias;
    l.NT = hidden <= 64 ? 8 : 16;                       // kernel template: gate tiles, padded
    const int NT = l.NT, H = hidden, nks = (in_dim + 1) / 2;
    size_t off = 0;
    l.wx_off = off; off += (size_t)nks * NT * 64;
    l.u_off = off; off += (size_t)(NT * 4) * NT * 64;
    l.bias_off = off; off += (size_t)NT * 32;
    l.wout_off = off; off += up16((size_t)NT * 8 * n_actions);
    l.bout_off = off; off += 16;
    l.flagw_off = off; off += (size_t)NT * 32;
    l.has_split = (in_dim == 2 && hidden == 128);
    if (l.has_split) {
        off = up16(off);
        l.uhi_off = off; off += (size_t)16 * 8 * 64 * 4;     // [16 tiles][8 k-steps][64 lanes] x 8 f16 = 4 floats
        l.ulo_off = off; off += (size_t)16 * 8 * 64 * 4;
    }
    l.has16 = (in_dim == 2 && hidden > 64);
    if (l.has16) {
        off = up16(off);
        l.u16_off = off; off += (size_t)32 * 32 * 64;
        l.wx16_off = off; off += (size_t)32 * 64;
        l.bias16_off = off; off += (size_t)32 * 16;
    }
    std::vector<float> pack(off, 0.0f);
    // gate-interleaved rows: row i of tile tau = (unit 8*tau + (i>>2), gate i&3) -> TF column gate*H + unit
    auto col = [&](int tau, int i) -> int {
        const int unit = 8 * tau + (i >> 2), gate = i & 3;
        return unit < H ? gate * H + unit : -1;
    };
    const size_t ld = 4 * (size_t)H;
    // Gate pre-activations leave the MFMAs already scaled for v_exp_f32 (2^x): i|f|o by -log2(e),
    // j by -2*log2(e); forget_bias joins the f bias (TF adds it before the sigmoid,
    // reference myasp-A1.py:177 LSTMCell(forget_bias=1.0)).
    const double L2E = 1.4426950408889634;
    auto gscale = [&](int i) -> double { return (i & 3) == 1 ? -2.0 * L2E : -L2E; };
    for (int ks = 0; ks < nks; ++ks)
        for (int tau = 0; tau < NT; ++tau)
            for (int ln = 0; ln < 64; ++ln) {
                const int feat = 2 * ks + (ln >> 5), cc = col(tau, ln & 31);
                pack[l.wx_off + ((size_t)ks * NT + tau) * 64 + ln] =
                    (feat < in_dim && cc >= 0) ? (float)(gscale(ln & 31) * kernel[(size_t)feat * ld + cc]) : 0.0f;
            }
    for (int kk = 0; kk < NT * 4; ++kk)
        for (int tau = 0; tau < NT; ++tau)
            for (int ln = 0; ln < 64; ++ln) {
                const int unit = 8 * (kk >> 2) + 2 * (kk & 3) + (ln >> 5), cc = col(tau, ln & 31);
                pack[l.u_off + ((size_t)kk * NT + tau) * 64 + ln] =
                    (unit < H && cc >= 0) ? (float)(gscale(ln & 31) * kernel[(size_t)(in_dim + unit) * ld + cc]) : 0.0f;
            }
    for (int tau = 0; tau < NT; ++tau)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int i = rowmap_h(r, half), cc = col(tau, i);
                const double fbias = (i & 3) == 2 ? (double)forget_bias : 0.0;
                pack[l.bias_off + (tau * 2 + half) * 16 + r] =
                    cc >= 0 ? (float)(gscale(i) * ((double)bias[cc] + fbias)) : 0.0f;
            }
    if (l.has_split) {
        // f16 fragments of U for the 3-term split kernel (lstm_f16x3.hip.h): lane (ha, row) of k-step s
        // holds, at i = 4*half' + q, the weight of hidden unit 8*(2s+ha) + 2q + half' into gate row `row`.
        _Float16* hi = reinterpret_cast<_Float16*>(pack.data() + l.uhi_off);
        _Float16* lo = reinterpret_cast<_Float16*>(pack.data() + l.ulo_off);
        for (int tau = 0; tau < 16; ++tau)
            for (int s8 = 0; s8 < 8; ++s8)
                for (int ln = 0; ln < 64; ++ln)
                    for (int i = 0; i < 8; ++i) {
                        const int row = ln & 31, ha = ln >> 5;
                        const int unit = 8 * (2 * s8 + ha) + 2 * (i & 3) + (i >> 2);
                        const int cc = col(tau, row);
                        const float wv = (float)(gscale(row) * kernel[(size_t)(in_dim + unit) * ld + cc]);
                        const _Float16 h16 = (_Float16)wv;
                        const size_t o = (((size_t)tau * 8 + s8) * 64 + ln) * 8 + i;
                        hi[o] = h16;
                        lo[o] = (_Float16)(wv - (float)h16);
                    }
    }
    if (l.has16) {
        // 16-window-tile layout (lstm_w16.hip.h): tile T row i = (unit 4T + (i>>2), gate i&3)
        auto col16 = [&](int T, int i) -> int {
            const int unit = 4 * T + (i >> 2), gate = i & 3;
            return unit < H ? gate * H + unit : -1;
        };
        for (int kk = 0; kk < 32; ++kk)
            for (int T = 0; T < 32; ++T)
                for (int ln = 0; ln < 64; ++ln) {
                    const int i = ln & 15, u = 4 * kk + (ln >> 4), cc = col16(T, i);
                    pack[l.u16_off + ((size_t)kk * 32 + T) * 64 + ln] =
                        (u < H && cc >= 0) ? (float)(gscale(i) * kernel[(size_t)(in_dim + u) * ld + cc]) : 0.0f;
                }
        for (int T = 0; T < 32; ++T)
            for (int ln = 0; ln < 64; ++ln) {
                const int i = ln & 15, feat = ln >> 4, cc = col16(T, i);
                pack[l.wx16_off + (size_t)T * 64 + ln] =
                    (feat < in_dim && cc >= 0) ? (float)(gscale(i) * kernel[(size_t)feat * ld + cc]) : 0.0f;
            }
        for (int T = 0; T < 32; ++T)
            for (int qq = 0; qq < 4; ++qq)
                for (int r = 0; r < 4; ++r) {
                    const int i = 4 * qq + r, cc = col16(T, i);
                    const double fbias = r == 2 ? (double)forget_bias : 0.0;
                    pack[l.bias16_off + ((size_t)T * 4 + qq) * 4 + r] =
                        cc >= 0 ? (float)(gscale(i) * ((double)bias[cc] + fbias)) : 0.0f;
                }
    }
    // the flag row of Wx (last input feature) in accumulator order, for the x-projection path
    for (int tau = 0; tau < NT; ++tau)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int i = rowmap_h(r, half), cc = col(tau, i);
                pack[l.flagw_off + (tau * 2 + half) * 16 + r] =
                    cc >= 0 ? (float)(gscale(i) * kernel[(size_t)(in_dim - 1) * ld + cc]) : 0.0f;
            }
    for (int u = 0; u < H; ++u)
        for (int a = 0; a < n_actions; ++a) pack[l.wout_off + (size_t)u * n_actions + a] = w_out[(size_t)u * n_actions + a];
    for (int a = 0; a < n_actions; ++a) pack[l.bout_off + a] = b_out[a];
    int rc = ensure(c, c->lstm_w, off * sizeof(float));
    if (rc) { c->lstm.set = false; return rc; }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->lstm_w.p, pack.data(), off * sizeof(float), hipMemcpyHostToDevice));
    l.set = true;
    c->lstm = l;
    return DRL_OK;
}

}  // extern "C"

// -------------------------------------------------------------------------------------------------
namespace {

struct OptDev {
    const float* in_scale = nullptr;
    const float* in_offset = nullptr;
    const float* rs = nullptr;      // RobustScaler table [4][P]: c_hi | c_lo | inv_hi | inv_lo
    int rsP = 0;
};

// The small host-side option arrays (per series / per flattened column) gathered into one block:
// in_scale[S] | in_offset[S] | RobustScaler table [4][P].
int opts_fill(drl_ctx* c, const drl_score_opts* o, int n_seg, int D, std::vector<float>& h, size_t& S, size_t& P) {
    if (o->in_scale && !o->in_offset) return fail(c, DRL_ERR_ARG, "opts.in_scale given without in_offset");
    if (o->vae_center && !o->vae_scale) return fail(c, DRL_ERR_ARG, "opts.vae_center given without vae_scale");
    S = o->in_scale ? up16(n_seg) : 0;
    P = o->vae_center ? up32(D) : 0;      // padded to a whole 32-feature tile (center 0, 1/scale 1):
    h.assign(2 * S + 4 * P, 0.0f);        // vae_wide16_kernel streams them per slab
    if (S) {
        memcpy(h.data(), o->in_scale, n_seg * sizeof(float));
        memcpy(h.data() + S, o->in_offset, n_seg * sizeof(float));
    }
    // center_ and 1/scale_ as float32 hi + lo pairs of the caller's float64 values (robust_apply, kernels.hip.h)
    float* hc = h.data() + 2 * S;
    for (size_t i = 0; i < P; ++i) {
        const bool in = i < (size_t)D;
        const double cen = in ? o->vae_center[i] : 0.0;
        const double scl = in ? o->vae_scale[i] : 1.0;
        if (in && !(std::isfinite(cen) && std::isfinite(scl) && scl != 0.0))
            return fail(c, DRL_ERR_ARG, "opts.vae_center/vae_scale[%zu] must be finite, scale non-zero", i);
        const double inv = 1.0 / scl;
        const float c_hi = (float)cen, i_hi = (float)inv;
        if (!std::isfinite(c_hi) || !std::isfinite(i_hi))
            return fail(c, DRL_ERR_ARG, "opts.vae_center/vae_scale[%zu] outside the float32 range", i);
        hc[i] = c_hi;
        hc[P + i] = (float)(cen - (double)c_hi);
        hc[2 * P + i] = i_hi;
        hc[3 * P + i] = (float)(inv - (double)i_hi);
    }
    return DRL_OK;
}
void opts_point(OptDev& d, const float* base, size_t S, size_t P) {
    if (S) { d.in_scale = base; d.in_offset = base + S; }
    if (P) { d.rs = base + 2 * S; d.rsP = (int)P; }
}

// Uploads that block into the context's growing buffer.  When it equals the one already on the
// device (the usual case: the same scaler for every call) nothing is copied, otherwise the stream is
// drained first — an earlier asynchronous call may still be reading the old block.
int upload_opts(drl_ctx* c, const drl_score_opts* o, int n_seg, int D, OptDev& d) {
    std::vector<float> h;
    size_t S = 0, P = 0;
    int rc = opts_fill(c, o, n_seg, D, h, S, P);
    if (rc) return rc;
    if (h.empty()) return DRL_OK;
    if (!(c->opt.p && c->opt_S == S && c->opt_P == P && c->opt_host.size() == h.size() &&
          memcmp(c->opt_host.data(), h.data(), h.size() * sizeof(float)) == 0)) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if ((rc = ensure(c, c->opt, h.size() * sizeof(float)))) return rc;
        HIP_TRY(c, hipMemcpy(c->opt.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        c->opt_host.swap(h);
        c->opt_S = S; c->opt_P = P;
    }
    opts_point(d, (const float*)c->opt.p, S, P);
    return DRL_OK;
}

// The same two tables for the entry point that allocates nothing: through the fixed pinned block
// into this GPU's scratch array, on the stream.  The block is reused by the next such call, which
// first waits for this copy (na_ev).
int noalloc_tables(drl_ctx* c, const int64_t* seg_offsets, int n_seg, int W, const drl_score_opts* o, int D,
                   SegHost& sh, OptDev& od) {
    if (c->na_stage_busy) { HIP_TRY(c, hipEventSynchronize(c->na_ev)); c->na_stage_busy = false; }
    int rc = seg_fill(c, seg_offsets, n_seg, W, false, c->na_stage, sh);
    if (rc) return rc;
    std::vector<float> h;                               // host memory only
    size_t S = 0, P = 0;
    if ((rc = opts_fill(c, o, n_seg, D, h, S, P))) return rc;
    const size_t ob = h.size() * sizeof(float);
    if (seg_bytes(n_seg) > DRL_SCRATCH_SEG_BYTES || ob > DRL_SCRATCH_OPT_BYTES)
        return fail(c, DRL_ERR_ARG, "drl_score_batch_dev_noalloc: tables exceed the scratch layout");   // unreachable within the stated limits
    if (ob) memcpy((char*)c->na_stage + DRL_SCRATCH_SEG_BYTES, h.data(), ob);
    HIP_TRY(c, hipMemcpyAsync(c->scratch_dev, c->na_stage, DRL_SCRATCH_SEG_BYTES + ob, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->na_ev, c->stream));
    c->na_stage_busy = true;
    seg_point(sh, c->scratch_dev, n_seg);
    opts_point(od, (const float*)(c->scratch_dev + DRL_SCRATCH_SEG_BYTES), S, P);
    return DRL_OK;
}

int check_opts(drl_ctx* c, const drl_score_opts* o, int F) {
    if (!o) return fail(c, DRL_ERR_ARG, "opts is NULL");
    if (o->struct_size != (int32_t)sizeof(drl_score_opts))
        return fail(c, DRL_ERR_ARG, "opts.struct_size %d != %zu", o->struct_size, sizeof(drl_score_opts));
    // revision 2: vae_center / vae_scale are const double* (revision 1, which left this field 0: const float*)
    if (o->abi_revision != DRL_ABI_REVISION)
        return fail(c, DRL_ERR_ARG, "opts.abi_revision %d: this library speaks revision %d of drl_score_opts "
                    "(set it to DRL_ABI_REVISION from the header you compile against)", o->abi_revision, DRL_ABI_REVISION);
    if (o->vae_lag != 0 && o->vae_lag != 1) return fail(c, DRL_ERR_ARG, "opts.vae_lag must be 0 or 1");
    if (o->flag_mode < 0 || o->flag_mode > 3) return fail(c, DRL_ERR_ARG, "opts.flag_mode %d", o->flag_mode);
    if (o->action_rule < 0 || o->action_rule > 1) return fail(c, DRL_ERR_ARG, "opts.action_rule %d", o->action_rule);
    if (o->flag_mode == DRL_FLAG_EXPLICIT && !o->flags) return fail(c, DRL_ERR_ARG, "explicit flag mode needs opts.flags");
    if (o->action_rule == DRL_ACTION_WADI && o->flag_mode != DRL_FLAG_BOTH)
        return fail(c, DRL_ERR_ARG, "the WaDi action rule needs DRL_FLAG_BOTH");
    (void)F;
    return DRL_OK;
}

// Device-pointer core shared by both entry points.
int score_impl(drl_ctx* c, const float* d_series, const int64_t* seg_offsets, int n_seg, int F, int W,
               const drl_score_opts* o, const float* d_flags, const float* d_eps,
               float* d_err, float* d_q, int32_t* d_action, float* d_margin, long long* K_out, bool noalloc = false,
               const OptDev* od_pre = nullptr) {
    // od_pre: option tables already on the device (the chunks of a pipelined host call share one upload)
    // noalloc (drl_score_batch_dev_noalloc): tables and projections live in this GPU's scratch array,
    // nothing is allocated, no profiling events are created
    const bool want_vae = d_err != nullptr;
    const bool want_q = d_q != nullptr || d_action != nullptr || d_margin != nullptr;
    if (want_vae) {
        if (!c->vae.set) return fail(c, DRL_ERR_STATE, "recon_err requested but drl_set_vae was not called");
        if (c->vae.D != W * F) return fail(c, DRL_ERR_ARG, "VAE input width %d != W*F = %d", c->vae.D, W * F);
        if (d_eps && (!c->vae.has_lv || c->vae.mu_idx < 1 || c->vae.latent > 32))
            return fail(c, DRL_ERR_UNSUPPORTED, "eps path needs a logvar layer, an encoder hidden layer and latent <= 32");
    }
    if (want_q) {
        if (!c->lstm.set) return fail(c, DRL_ERR_STATE, "Q requested but drl_set_lstm was not called");
        // in_dim == F: a network without the flag channel (the reference's legacy RLVAL / AdaptiveTauRL
        // estimators take the value alone); only DRL_FLAG_ZEROS makes sense there
        if (c->lstm.I == F && o->flag_mode != DRL_FLAG_ZEROS)
            return fail(c, DRL_ERR_ARG, "LSTM in_dim %d == F: the network has no flag channel, use DRL_FLAG_ZEROS", c->lstm.I);
        if (c->lstm.I != F + 1 && c->lstm.I != F)
            return fail(c, DRL_ERR_ARG, "LSTM in_dim %d != F+1 = %d (or F, for a network without the flag channel)", c->lstm.I, F + 1);
        if (!d_q) return fail(c, DRL_ERR_ARG, "action/margin need the q output buffer");
    }
    SegHost sh;
    OptDev od;
    int rc;
    // Whatever path leaves this function after kernels may have been queued (an error after a launch included),
    // the table slot's event is recorded behind them: an eviction of the slot then waits for THOSE kernels, not
    // for an older call's.
    struct SlotGuard {
        drl_ctx* c; bool on;
        ~SlotGuard() { if (on && seg_mark(c) != DRL_OK) (void)hipGetLastError(); }
    } guard{c, false};
    if (noalloc) {
        c->seg_last = -1;
        if ((rc = noalloc_tables(c, seg_offsets, n_seg, W, o, W * F, sh, od))) return rc;
        if (K_out) *K_out = sh.K_total;
        if (sh.K_total == 0) return DRL_OK;
    } else {
        if ((rc = build_segments(c, seg_offsets, n_seg, W, false, sh))) return rc;
        if (K_out) *K_out = sh.K_total;
        if (sh.K_total == 0) return DRL_OK;
        guard.on = true;
        if (od_pre) od = *od_pre;
        else if ((rc = upload_opts(c, o, n_seg, W * F, od))) return rc;
    }
    // per-sample input projections of the xproj paths: the context's growing buffer, or the scratch array
    auto projections = [&](size_t bytes, float** P) -> int {
        if (noalloc) {
            if (bytes > DRL_SCRATCH_BYTES - DRL_SCRATCH_P_OFFSET)
                return fail(c, DRL_ERR_ARG, "drl_score_batch_dev_noalloc: %lld samples need %zu bytes of projections, "
                            "the scratch array holds %u (limit: %u samples at hidden <= 64, %u above)",
                            (long long)seg_offsets[n_seg], bytes, DRL_SCRATCH_BYTES - DRL_SCRATCH_P_OFFSET,
                            (DRL_SCRATCH_BYTES - DRL_SCRATCH_P_OFFSET) / 1024, (DRL_SCRATCH_BYTES - DRL_SCRATCH_P_OFFSET) / 2048);
            *P = (float*)(c->scratch_dev + DRL_SCRATCH_P_OFFSET);
            return DRL_OK;
        }
        if (c->pbuf_max >= 0 && bytes > (size_t)c->pbuf_max)       // tests: behave as if a larger allocation had failed
            return fail(c, DRL_ERR_HIP, "projections: %zu bytes over the pbuf_max knob", bytes);
        int r = ensure(c, c->pbuf, bytes);
        if (r) return r;
        *P = (float*)c->pbuf.p;
        return DRL_OK;
    };
    const bool prof = c->prof.on && !noalloc;

    SegTab seg;
    seg.samp_off = sh.d_samp; seg.win_off = sh.d_win; seg.tile_off = sh.d_t32;
    seg.in_scale = od.in_scale; seg.in_offset = od.in_offset; seg.n_seg = n_seg; seg.W = W;

    // Wide windows, more 64-window tiles than CUs: the ping-pong VAE kernel (vae_wide16pp.hip.h).  Its persistent
    // workgroups want every CU for the whole launch, so such a call keeps its VAE kernel on the main stream.
    const bool pp_wanted = want_vae && c->vae.wide16 && !d_eps && c->vae_pp != 0 && (((uintptr_t)d_series) & 15) == 0 &&
                           (c->vae_pp == 2 || sh.tiles64 > c->n_cu);
    auto launch_vae = [&](hipStream_t vst) -> int {
        VaeArgs a;
        memset(&a, 0, sizeof a);
        fill_vae_args(c, a);
        a.x = d_series; a.win_stride = F; a.lag_elems = (1 - o->vae_lag) * F; a.D = W * F;
        a.rs = od.rs; a.rsP = od.rsP; a.clip = o->vae_clip;
        a.eps = d_eps; a.seg = seg; a.n_tiles = sh.tiles32; a.err = d_err; a.recon = nullptr;
        // wide windows: stream the first/last layer through LDS when the workgroup's flat region fits
        const int xr_floats = (63 * F + ((W * F + 31) & ~31) + 3) & ~3;
        size_t wide_lds = ((size_t)2 * VW16_SLAB + 2 * VW16_SIDE + (size_t)xr_floats) * sizeof(float);
        if (c->vw16_pad) wide_lds = std::min<size_t>(wide_lds + (size_t)c->vw16_pad, 160 * 1024);   // measurement knob: fewer workgroups per CU
        // vae_wide16s_kernel wherever one of its workgroups fits: two per CU for the benchmark shapes (C4 2.41 ms against
        // vae_wide16_kernel's 2.66), one per CU for wider inputs (F = 140: 2.30 ms against 2.78; a ring of three slabs
        // with one barrier per slab measured the same 2.31 there, so it is not instantiated)
        const size_t s_lds = ((size_t)2 * VW16_SLAB + 2 * VW16_SIDE + (size_t)xr_floats) * sizeof(float);
        const bool use_s = c->vae_s != 0 && s_lds <= 160 * 1024;
        // with eps (the reference's sampled z): only vae_wide16s_kernel carries the logvar head among the wide kernels
        const bool wide = c->vae.wide16 && wide_lds <= 160 * 1024 && (!d_eps || (use_s && c->vae.latent <= 32));
        // the ping-pong form: two regions (+ up to 3 floats each for the 16-byte alignment of their loads) and one ring
        const int xr_pp = xr_floats + 4;
        const size_t pp_lds = ((size_t)2 * VW16_SLAB + 2 * VW16_SIDE + (size_t)2 * xr_pp) * sizeof(float);
        const bool use_pp = wide && pp_wanted && pp_lds <= 160 * 1024;
        if (prof) prof_begin(c, 1, vst);
        if (wide) {
            if (wide_lds > c->wide_lds_set) {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16s_kernel<true, 2, 8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16s_kernel<false, 2, 8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16s_kernel<true, 2, 4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16s_kernel<false, 2, 4>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16pp_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(vae_wide16pp_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
                c->wide_lds_set = 160 * 1024;
            }
            Vae16Args b16;
            memset(&b16, 0, sizeof b16);
            for (int p = 0; p < c->vae.n_path; ++p) {
                b16.w16[p] = (const float*)c->vae_w.p + c->vae.w16_off[p];
                b16.b16[p] = (const float*)c->vae_w.p + c->vae.b16_off[p];
            }
            {
                // the middle layers' weights and biases are contiguous in the pack: staged into the ring when they fit
                const int lastp = c->vae.n_path - 1;
                const size_t span = lastp > 1 ? c->vae.w16_off[lastp] - c->vae.w16_off[1] : 0;
                b16.mid_floats = (span > 0 && span <= 2 * VW16_SLAB && span % 4 == 0) ? (int)span : 0;
            }
            if (d_eps) {
                b16.wlv16 = (const float*)c->vae_w.p + c->vae.wlv16_off;
                b16.blv16 = (const float*)c->vae_w.p + c->vae.blv16_off;
            }
            a.seg.tile_off = sh.d_t64;               // a tile = 64 windows of one series
            a.n_tiles = sh.tiles64;
            if (use_pp && !d_eps) {
                // two tiles per workgroup in ping-pong, persistent: every CU a balanced share of the tiles
                const dim3 grid(std::min(c->n_cu, (sh.tiles64 + 1) / 2));
                if (a.rs) hipLaunchKernelGGL(vae_wide16pp_kernel<true>, grid, dim3(VWP_THREADS), pp_lds, vst, a, b16, xr_pp);
                else hipLaunchKernelGGL(vae_wide16pp_kernel<false>, grid, dim3(VWP_THREADS), pp_lds, vst, a, b16, xr_pp);
                c->last_vae_variant = "wide16pp";
            } else if (use_s) {
                const dim3 grid(sh.tiles64);         // a workgroup = one tile, one workgroup per CU
                // hidden layers of at most 64 units beside the windows: the instantiation that skips the zero row tiles
                const bool nt4 = c->vae.dout[0] <= 64 && c->vae.din[c->vae.n_path - 1] <= 64;
                if (nt4) {
                    if (a.rs) hipLaunchKernelGGL((vae_wide16s_kernel<true, 2, 4>), grid, dim3(256), s_lds, vst, a, b16, xr_floats);
                    else hipLaunchKernelGGL((vae_wide16s_kernel<false, 2, 4>), grid, dim3(256), s_lds, vst, a, b16, xr_floats);
                } else {
                    if (a.rs) hipLaunchKernelGGL((vae_wide16s_kernel<true, 2, 8>), grid, dim3(256), s_lds, vst, a, b16, xr_floats);
                    else hipLaunchKernelGGL((vae_wide16s_kernel<false, 2, 8>), grid, dim3(256), s_lds, vst, a, b16, xr_floats);
                }
                c->last_vae_variant = "wide16s";
            } else {
                const dim3 grid(sh.tiles64);         // a workgroup = one tile
                if (a.rs) hipLaunchKernelGGL(vae_wide16_kernel<true>, grid, dim3(256), wide_lds, vst, a, b16, xr_floats);
                else hipLaunchKernelGGL(vae_wide16_kernel<false>, grid, dim3(256), wide_lds, vst, a, b16, xr_floats);
                c->last_vae_variant = "wide16";
            }
        } else {
            int rcl = launch_vae_chain(c, a, sh.tiles32, vst);
            if (rcl) return rcl;
        }
        if (prof) prof_end(c, vst);
        HIP_TRY(c, hipGetLastError());
        return DRL_OK;
    };
    // Worth it when the call is large — measured (round 2) +3 % on C4 (172,776 windows, D = 3075), +2 % on C2
    // (93,191 windows: the 0.05 ms of the chain VAE kernel disappear under the LSTM kernel's last round),
    // +0.3 % on C5, but -3 % on C3 (28,454 windows: the LSTM grid is a single round that wants every CU
    // at once) — and when the call is so small that the LSTM grid leaves CUs idle anyway (C1, 1,395
    // windows: +13 %).  The second stream's priority makes no difference (DRLVAE_AUX_PRIO, tools/ab_prio.sh).
    // DRLVAE_OVERLAP=0 turns it off, =2 forces it for every call.
    const bool big_wide = sh.K_total >= 65536;
    const bool small = (long long)sh.tiles32 * (o->flag_mode == DRL_FLAG_BOTH ? 2 : 1) <= (long long)c->n_cu * 3 / 4;
    const bool overlap = want_vae && want_q && c->aux_stream &&
                         (c->overlap_mode == 2 || (c->overlap_mode == 1 && ((big_wide && !pp_wanted) || small)));
    c->last_overlap = overlap;
    if (want_vae && !overlap && (rc = launch_vae(c->stream))) return rc;
    if (overlap) {
        HIP_TRY(c, hipEventRecord(c->ev_fork, c->stream));
        HIP_TRY(c, hipStreamWaitEvent(c->aux_stream, c->ev_fork, 0));
    }
    if (want_q) {
        LstmArgs a;
        memset(&a, 0, sizeof a);
        fill_lstm_args(c, a);
        a.W = W; a.x = d_series; a.F = F; a.win_stride = F; a.row_stride = F; a.row0 = 1;
        a.flag_mode = o->flag_mode; a.flags = d_flags; a.seg = seg;
        a.S = o->flag_mode == DRL_FLAG_BOTH ? 2 : 1;
        a.q = d_q;
        // argmax epilogue inside the LSTM kernel whenever the workgroup that computes a window's Q holds every
        // value the rule needs: one candidate state, or both candidates in one wave (lstm_rec2_kernel for
        // every tile).  Otherwise finalize_kernel below.  DRLVAE_NO_FUSED_EPILOGUE=1: always finalize_kernel (tests).
        bool fused_epilogue = (d_action || d_margin) && a.S == 1 && !c->no_fused_epilogue;
        if (fused_epilogue) { a.action = d_action; a.margin = d_margin; a.rule = o->action_rule; }
        // hidden widths 65..128 are packed zero-padded to 16 gate tiles x 64 k-steps (padded units stay at h = 0)
        const bool fast = c->lstm.I == 2 && c->lstm.NT == 16 && F == 1 && W >= 2 && W + 65 <= H128_SV;
        // the other shapes project every sample once (1 KiB per sample for H <= 64, 2 KiB above: C4 177 MB).  When
        // that buffer cannot be had — beyond 64 GiB, or the allocation fails — the call still runs, on the fallback
        // kernel that re-projects every window's rows (the no-allocation entry point reports its limit instead).
        float* Pbuf = nullptr;
        bool use_proj = false;
        if (!fast) {
            const size_t per_sample = (size_t)c->lstm.NT * 32 * sizeof(float);
            if ((size_t)seg_offsets[n_seg] * per_sample <= ((size_t)64 << 30)) {
                rc = projections((size_t)seg_offsets[n_seg] * per_sample, &Pbuf);
                if (rc == DRL_OK) use_proj = true;
                else if (noalloc) return rc;
                else (void)hipGetLastError();               // out of memory for the projections: fall back
            }
        }
        c->last_tile = 0;
        c->last_kernels.clear();
        auto ran = [&](const char* name) { if (!c->last_kernels.empty()) c->last_kernels += ';'; c->last_kernels += name; };
        // Persistent kernels whose G workgroups (per candidate state) walk 64-window tiles: when the last round is at most
        // half full, its tiles run as 32-window tiles in a second launch of the <1> instantiation — half as long, one tile
        // per workgroup (C3 with 128 hidden units: 445 tiles on 128 workgroups = 3.48 rounds, 3.5 this way instead of 4).
        // A 64-window tile j of a series is its 32-window tiles 2j and 2j+1, so the 64-window tiles from n_main on are the
        // 32-window tiles from t0 on.  n_main = tiles64 when the call is not split.
        auto half_tile_split = [&](int G, int& n_main, int& t0) {
            n_main = sh.tiles64; t0 = sh.tiles32;
            const int rem = sh.tiles64 % G;
            // (worth a second launch — its own weight load and a gap on the stream — only while half a round is a few per cent
        // of the call: up to 16 rounds; C5's 244 rounds measure the same end to end and 0.7 % more kernel span with it)
        if (!c->h128_split || sh.tiles64 <= G || sh.tiles64 >= 16 * G || rem == 0 || 2 * rem > G) return;
            n_main = sh.tiles64 - rem;
            long long a64 = 0, a32 = 0;
            for (int i = 0; i < n_seg; ++i) {
                const long long K = std::max<long long>(0, seg_offsets[i + 1] - seg_offsets[i] - W);
                const long long s64 = (K + 63) / 64;
                if (n_main < a64 + s64) { t0 = (int)(a32 + 2 * (n_main - a64)); break; }
                a64 += s64; a32 += (K + 31) / 32;
            }
        };
        if (prof) prof_begin(c, 0, c->stream, want_vae && !overlap);   // serial: the VAE kernel's end event is the start
        if (fast && c->tl_n > 0 && c->lstm.has16) {
            // a sparse pass of the rollout's fixed point: only the listed 16-window tiles
            a.seg.tile_off = sh.d_t16; a.n_tiles = c->tl_n; a.tile_list = c->tl_dev;
            Lstm16Args fa;
            fa.l = a;
            fa.u16 = (const float*)c->lstm_w.p + c->lstm.u16_off;
            fa.wx16 = (const float*)c->lstm_w.p + c->lstm.wx16_off;
            fa.bias16 = (const float*)c->lstm_w.p + c->lstm.bias16_off;
            // more tiles than CUs: a workgroup walks several (the recurrent weights are loaded once)
            hipLaunchKernelGGL(lstm_h128i2_w16_kernel, dim3(std::min(c->tl_n, c->n_cu), a.S), dim3(256), 0, c->stream, fa);
            ran("lstm_h128i2_w16_kernel");
            c->last_variant = "h128i2"; c->last_tile = 16;
        } else if (fast && c->lstm_math == 1 && c->lstm.has_split) {
            a.seg.tile_off = sh.d_t64; a.n_tiles = sh.tiles64;
            LstmF16Args fa;
            fa.l = a;
            fa.uhi = reinterpret_cast<const u32x4*>((const float*)c->lstm_w.p + c->lstm.uhi_off);
            fa.ulo = reinterpret_cast<const u32x4*>((const float*)c->lstm_w.p + c->lstm.ulo_off);
            hipLaunchKernelGGL(lstm_h128i2_f16x3_kernel, dim3(std::min<unsigned>(sh.tiles64, c->n_cu), a.S), dim3(256), 0, c->stream, fa);
            ran("lstm_h128i2_f16x3_kernel");
            c->last_variant = "h128i2_f16x3"; c->last_tile = 64;
        } else if (fast && c->lstm.has16 && sh.tiles16 * a.S <= c->n_cu) {
            // a call of a few thousand windows (one series): 16-window tiles, twice as many workgroups again
            a.seg.tile_off = sh.d_t16; a.n_tiles = sh.tiles16;
            Lstm16Args fa;
            fa.l = a;
            fa.u16 = (const float*)c->lstm_w.p + c->lstm.u16_off;
            fa.wx16 = (const float*)c->lstm_w.p + c->lstm.wx16_off;
            fa.bias16 = (const float*)c->lstm_w.p + c->lstm.bias16_off;
            hipLaunchKernelGGL(lstm_h128i2_w16_kernel, dim3(sh.tiles16, a.S), dim3(256), 0, c->stream, fa);
            ran("lstm_h128i2_w16_kernel");
            c->last_variant = "h128i2"; c->last_tile = 16;
        } else if (fast && sh.tiles32 <= c->n_cu) {
            // a small call (one series, a replay batch): 32-window tiles over twice as many CUs
            a.seg.tile_off = sh.d_t32; a.n_tiles = sh.tiles32;
            hipLaunchKernelGGL(lstm_h128i2_kernel<1>, dim3(sh.tiles32, a.S), dim3(256), 0, c->stream, a);
            ran("lstm_h128i2_kernel<1>");
            c->last_variant = "h128i2"; c->last_tile = 32;
        } else if (fast) {
            // persistent: one workgroup per CU walks the tiles (the recurrent weights are loaded once).  A last round
            // that is at most half full runs as 32-window tiles in a second launch (see half_tile_split)
            int n_main, t0;
            half_tile_split(c->n_cu, n_main, t0);
            a.seg.tile_off = sh.d_t64; a.n_tiles = n_main;
            hipLaunchKernelGGL(lstm_h128i2_kernel<2>, dim3(std::min<unsigned>(n_main, c->n_cu), a.S), dim3(256), 0, c->stream, a);
            ran("lstm_h128i2_kernel<2>");
            if (n_main < sh.tiles64) {
                a.seg.tile_off = sh.d_t32; a.n_tiles = sh.tiles32; a.tile0 = t0;
                hipLaunchKernelGGL(lstm_h128i2_kernel<1>, dim3(std::min<unsigned>(sh.tiles32 - t0, c->n_cu), a.S), dim3(256), 0, c->stream, a);
                ran("lstm_h128i2_kernel<1>");
                a.tile0 = 0; a.seg.tile_off = sh.d_t64; a.n_tiles = sh.tiles64;
            }
            c->last_variant = "h128i2"; c->last_tile = 64;
        } else if (c->lstm.NT == 16 && use_proj) {
            // multivariate H = 128: project every sample once (16 gate tiles), then the register-stationary
            // recurrence from the projections
            const long long n_samples = seg_offsets[n_seg];
            XprojArgs xa;
            xa.x = d_series; xa.n_samples = n_samples; xa.F = F; xa.wx = a.wx; xa.bias = a.bias;
            xa.P = Pbuf; xa.nt_total = 16;
            xa.samp_off = sh.d_samp; xa.n_seg = n_seg; xa.in_scale = od.in_scale; xa.in_offset = od.in_offset;
            hipLaunchKernelGGL(xproj_kernel<8>, dim3((unsigned)((n_samples + 127) / 128), 2), dim3(256), 0, c->stream, xa);
            ran("xproj_kernel<8>");
            RecArgs ra;
            ra.P = xa.P; ra.n_samples = n_samples; ra.win_rows = 1;
            ra.flagw = (const float*)c->lstm_w.p + c->lstm.flagw_off;
            if (sh.tiles32 * a.S <= c->n_cu) {
                a.seg.tile_off = sh.d_t32; a.n_tiles = sh.tiles32;
                ra.l = a;
                hipLaunchKernelGGL(lstm_h128p_kernel<1>, dim3(sh.tiles32, a.S), dim3(256), 0, c->stream, ra);
                ran("lstm_h128p_kernel<1>");
            } else {
                const int G = std::max(1, c->n_cu / a.S);
                int n_main, t0;
                half_tile_split(G, n_main, t0);
                a.seg.tile_off = sh.d_t64; a.n_tiles = n_main;
                ra.l = a;
                const unsigned gx = std::min<unsigned>(n_main, G);
                hipLaunchKernelGGL(lstm_h128p_kernel<2>, dim3(gx, a.S), dim3(256), 0, c->stream, ra);
                ran("lstm_h128p_kernel<2>");
                if (n_main < sh.tiles64) {
                    a.seg.tile_off = sh.d_t32; a.n_tiles = sh.tiles32;
                    ra.l = a; ra.tile0 = t0;
                    hipLaunchKernelGGL(lstm_h128p_kernel<1>, dim3(std::min<unsigned>(sh.tiles32 - t0, G), a.S), dim3(256), 0, c->stream, ra);
                    ran("lstm_h128p_kernel<1>");
                    a.seg.tile_off = sh.d_t64; a.n_tiles = sh.tiles64;
                }
            }
            c->last_variant = "xproj+h128";
        } else if (c->lstm.NT == 8 && use_proj) {
            // H <= 64: project every sample once, then run the recurrences from the projections (a chunk of P stays < 4 GiB)
            const long long n_samples = seg_offsets[n_seg];
            size_t lds = ((size_t)32 * 8 * 64 + 8 * 32) * sizeof(float);   // U + the flag row
            if (c->rec_lds_pad) lds = std::min<size_t>(lds + (size_t)c->rec_lds_pad, 160 * 1024);   // measurement knob: workgroups per CU
            if (!c->rec_attr_set) {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec_kernel<8, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec_kernel<8, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec2_kernel<8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                c->rec_attr_set = true;
            }
            XprojArgs xa;
            xa.x = d_series; xa.n_samples = n_samples; xa.F = F; xa.wx = a.wx; xa.bias = a.bias;
            xa.P = Pbuf; xa.nt_total = 8;
            xa.samp_off = sh.d_samp; xa.n_seg = n_seg; xa.in_scale = od.in_scale; xa.in_offset = od.in_offset;
            hipLaunchKernelGGL(xproj_kernel<8>, dim3((unsigned)((n_samples + 127) / 128)), dim3(256), 0, c->stream, xa);
            ran("xproj_kernel<8>");
            RecArgs ra;
            a.n_tiles = sh.tiles32;
            ra.l = a; ra.P = xa.P; ra.n_samples = n_samples; ra.win_rows = 1;
            ra.flagw = (const float*)c->lstm_w.p + c->lstm.flagw_off;
            const unsigned rec_grid = (sh.tiles32 + 3) / 4;       // a workgroup is 4 waves, a wave one tile of 32 windows
            if (a.S == 2) {
                // Two candidate states.  lstm_rec2_kernel runs both in one wave (shared projections and weight
                // fragments: ~5 % less time per window) but is twice as long per wave; lstm_rec_kernel gives
                // each candidate its own wave.  With `slots` waves on the chip at a time, whole rounds of
                // `slots` tiles go to the first, and a remainder that fits the chip as one wave per
                // (tile, candidate) goes to the second — one round of half the length instead of a full one.
                const int slots = 4 * c->n_cu;
                const int rem = sh.tiles32 % slots;
                // (Measured, one MI355X: C3 0.545 -> 0.517 ms, C4 2.83 -> 2.74 ms on one stream.  When the VAE
                // kernel runs beside the LSTM on the second stream the CUs are shared between the two grids
                // and the rounds no longer exist: C4 measured 29.7 M windows/s with one wave per candidate,
                // 29.5 M with lstm_rec2_kernel for every tile, 28.8 M split — so those calls keep the former.)
                const int n2 = c->rec2_mode == 0 ? 0 : c->rec2_mode == 2 ? sh.tiles32
                               : overlap ? 0
                               : 2 * rem <= slots ? sh.tiles32 - rem : sh.tiles32;
                if (n2 == sh.tiles32 && (d_action || d_margin) && !c->no_fused_epilogue) {
                    fused_epilogue = true;
                    ra.l.action = d_action; ra.l.margin = d_margin; ra.l.rule = o->action_rule;
                }
                if (n2 > 0) {
                    ra.l.n_tiles = n2; ra.tile0 = 0;
                    hipLaunchKernelGGL(lstm_rec2_kernel<8>, dim3((n2 + 3) / 4), dim3(256), lds, c->stream, ra);
                    ran("lstm_rec2_kernel<8>");
                }
                if (n2 < sh.tiles32) {
                    ra.l.n_tiles = sh.tiles32; ra.tile0 = n2;
                    hipLaunchKernelGGL((lstm_rec_kernel<8, true>), dim3((sh.tiles32 - n2 + 1) / 2), dim3(256), lds, c->stream, ra);
                    ran("lstm_rec_kernel<8, true>");
                }
            } else if (a.flag_mode > 0) {
                hipLaunchKernelGGL((lstm_rec_kernel<8, true>), dim3(rec_grid), dim3(256), lds, c->stream, ra);
                ran("lstm_rec_kernel<8, true>");
            } else {
                hipLaunchKernelGGL((lstm_rec_kernel<8, false>), dim3(rec_grid), dim3(256), lds, c->stream, ra);
                ran("lstm_rec_kernel<8, false>");
            }
            c->last_variant = "xproj+rec";
        } else {
            a.n_tiles = sh.tiles32;
            const dim3 grid((sh.tiles32 + 3) / 4, a.S);
            if (c->lstm.NT == 8) hipLaunchKernelGGL(lstm_generic_kernel<8>, grid, dim3(256), 0, c->stream, a);
            else hipLaunchKernelGGL(lstm_generic_kernel<16>, grid, dim3(256), 0, c->stream, a);
            ran(c->lstm.NT == 8 ? "lstm_generic_kernel<8>" : "lstm_generic_kernel<16>");
            c->last_variant = "generic";
        }
        if (prof) prof_end(c, c->stream);
        HIP_TRY(c, hipGetLastError());
        if (overlap) {
            // the LSTM grid is already queued (it is the long one and takes the CUs first)
            if ((rc = launch_vae(c->aux_stream))) return rc;
            HIP_TRY(c, hipEventRecord(c->ev_join, c->aux_stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
        if ((d_action || d_margin) && !fused_epilogue) {
            const int bs = 256;
            hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((sh.K_total + bs - 1) / bs)), dim3(bs), 0, c->stream,
                               d_q, sh.K_total, a.S, c->lstm.A, o->action_rule, d_action, d_margin);
            HIP_TRY(c, hipGetLastError());
        }
    }
    if (noalloc) return DRL_OK;
    guard.on = false;
    return seg_mark(c);
}

long long count_windows(const int64_t* seg_offsets, int n_seg, int W) {
    long long k = 0;
    for (int i = 0; i < n_seg; ++i) k += std::max<long long>(0, seg_offsets[i + 1] - seg_offsets[i] - W);
    return k;
}

}  // namespace

extern "C" {

int drl_score_batch_dev(drl_ctx* c, const float* series, const int64_t* seg_offsets, int n_seg,
                        int F, int W, const drl_score_opts* o,
                        float* recon_err, float* q, int32_t* action, float* margin) {
    if (!c) return DRL_ERR_ARG;
    if (!series || !seg_offsets || n_seg < 1 || F < 1 || W < 1)
        return fail(c, DRL_ERR_ARG, "drl_score_batch_dev: bad series/seg_offsets/n_seg/F/W");
    int rc = check_opts(c, o, F);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    return score_impl(c, series, seg_offsets, n_seg, F, W, o, o->flags, o->eps, recon_err, q, action, margin, nullptr);
}

int drl_score_batch_dev_noalloc(drl_ctx* c, const float* series, const int64_t* seg_offsets, int n_seg,
                                int F, int W, const drl_score_opts* o,
                                float* recon_err, float* q, int32_t* action, float* margin) {
    if (!c) return DRL_ERR_ARG;
    if (!series || !seg_offsets || n_seg < 1 || F < 1 || W < 1)
        return fail(c, DRL_ERR_ARG, "drl_score_batch_dev_noalloc: bad series/seg_offsets/n_seg/F/W");
    int rc = check_opts(c, o, F);
    if (rc) return rc;
    if (!c->scratch_dev || !c->na_stage || !c->na_ev)
        return fail(c, DRL_ERR_STATE, "drl_score_batch_dev_noalloc: the scratch array was not set up at drl_create");
    if (n_seg > DRL_NOALLOC_MAX_SEG)
        return fail(c, DRL_ERR_ARG, "drl_score_batch_dev_noalloc: %d series, the limit is %d", n_seg, DRL_NOALLOC_MAX_SEG);
    if ((long long)W * F > DRL_NOALLOC_MAX_D)
        return fail(c, DRL_ERR_ARG, "drl_score_batch_dev_noalloc: window of %lld values, the limit is %d", (long long)W * F, DRL_NOALLOC_MAX_D);
    HIP_TRY(c, hipSetDevice(c->device));
    return score_impl(c, series, seg_offsets, n_seg, F, W, o, o->flags, o->eps, recon_err, q, action, margin, nullptr, true);
}

int drl_score_batch(drl_ctx* c, const float* series, const int64_t* seg_offsets, int n_seg,
                    int F, int W, const drl_score_opts* o,
                    float* recon_err, float* q, int32_t* action, float* margin) {
    if (!c) return DRL_ERR_ARG;
    if (!series || !seg_offsets || n_seg < 1 || F < 1 || W < 1)
        return fail(c, DRL_ERR_ARG, "drl_score_batch: bad series/seg_offsets/n_seg/F/W");
    int rc = check_opts(c, o, F);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const long long total = seg_offsets[n_seg] - seg_offsets[0];
    if (seg_offsets[0] != 0) return fail(c, DRL_ERR_ARG, "seg_offsets[0] must be 0");
    for (int i = 0; i < n_seg; ++i)
        if (seg_offsets[i + 1] < seg_offsets[i]) return fail(c, DRL_ERR_ARG, "seg_offsets must be non-decreasing (series %d)", i);
    const long long K = count_windows(seg_offsets, n_seg, W);
    if (K == 0) return DRL_OK;
    const int S = o->flag_mode == DRL_FLAG_BOTH ? 2 : 1;
    const int A = c->lstm.set ? c->lstm.A : 2;
    const bool want_q = q || action || margin;
    // action/margin without q: the q block is staged on the device all the same (want_q)
    if ((rc = ensure(c, c->st_in, (size_t)total * F * sizeof(float)))) return rc;
    const bool explicit_flags = o->flag_mode == DRL_FLAG_EXPLICIT;
    if (explicit_flags && (rc = ensure(c, c->st_flags, (size_t)total * sizeof(float)))) return rc;
    // an error return after the first asynchronous copy waits for the copies still reading the caller's
    // buffers (a pinned host buffer is read by the DMA engine after hipMemcpyAsync has returned)
    auto drain = [&](int code) {
        if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        return code;
    };
    const float* d_eps = nullptr;
    if (o->eps && recon_err) {
        if (!c->vae.set) return fail(c, DRL_ERR_STATE, "eps given but drl_set_vae was not called");
        const size_t n = (size_t)K * c->vae.latent * sizeof(float);
        if ((rc = ensure(c, c->st_eps, n))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->st_eps.p, o->eps, n, hipMemcpyHostToDevice, c->stream));
        d_eps = (const float*)c->st_eps.p;
    }
    if (recon_err && (rc = ensure(c, c->st_err, (size_t)K * sizeof(float)))) return drain(rc);
    if (want_q && (rc = ensure(c, c->st_q, (size_t)K * S * A * sizeof(float)))) return drain(rc);
    if (action && (rc = ensure(c, c->st_act, (size_t)K * sizeof(int32_t)))) return drain(rc);
    if (margin && (rc = ensure(c, c->st_margin, (size_t)K * sizeof(float)))) return drain(rc);
    float* const d_in = (float*)c->st_in.p;
    float* const d_fl = explicit_flags ? (float*)c->st_flags.p : nullptr;

    // Large inputs (WaDi: 85 MB of sensors per call) are pipelined: the series is cut into chunks of whole
    // series and window ranges of long ones (a range carries its W-sample halo, which is already on the
    // device from the chunk before); chunk i+1 travels on the copy stream while the kernels of chunk i run.
    // A copy from pageable memory keeps this thread busy but not the GPU.  Windows are independent, so the
    // chunks' results are those of the one-shot call, bit for bit (tested).  The get_states_list flag quirk
    // depends on a window's position in its series, so series are not cut in that mode.
    const size_t in_bytes = (size_t)total * F * sizeof(float);
    size_t PIPE_MIN = (size_t)16 << 20;
    const size_t CHUNK = (size_t)8 << 20;
    if (c->pipe_min > 0) PIPE_MIN = (size_t)c->pipe_min;   // tests: pipeline small inputs too
    const bool pipelined = in_bytes >= PIPE_MIN && c->copy_stream && c->copy_ev[0] && !c->no_pipeline;
    if (!pipelined) {
        HIP_TRY(c, hipMemcpyAsync(d_in, series, in_bytes, hipMemcpyHostToDevice, c->stream));
        if (explicit_flags)
            HIP_TRY(c, hipMemcpyAsync(d_fl, o->flags, (size_t)total * sizeof(float), hipMemcpyHostToDevice, c->stream));
        rc = score_impl(c, d_in, seg_offsets, n_seg, F, W, o, d_fl, d_eps,
                        recon_err ? (float*)c->st_err.p : nullptr, want_q ? (float*)c->st_q.p : nullptr,
                        action ? (int32_t*)c->st_act.p : nullptr, margin ? (float*)c->st_margin.p : nullptr, nullptr);
        if (rc) return drain(rc);
    } else {
        OptDev od_all;                                   // one upload of the option tables for every chunk
        if ((rc = upload_opts(c, o, n_seg, W * F, od_all))) return drain(rc);
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // st_in may still be read by an earlier asynchronous call
        // Chunk sizes grow: a first chunk of 8 MB puts the GPU to work after a third of a millisecond, the later
        // ones are large enough to fill the chip (1,024 waves of 32 windows) — a call costs the
        // kernels' time plus that first copy, not plus the whole upload (C4: 8.9 -> 7.0 ms, tools/host_path_rate.py).
        long long chunk_samples = std::max<long long>((long long)(CHUNK / (F * sizeof(float))), 4 * (long long)W);
        long long chunk_cap = std::max<long long>(chunk_samples, 32768 + (long long)W);
        if (c->pipe_chunk > 0)                                    // tests, tuning: fixed chunk size in samples
            chunk_samples = chunk_cap = std::max<long long>(c->pipe_chunk, 4 * (long long)W);
        const bool may_cut = o->flag_mode != DRL_FLAG_QUIRK;
        long long copied = 0;                            // samples [0, copied) are on the device or on their way
        long long win_done = 0;                          // windows of the chunks before this one
        int ser = 0;                                     // current series
        long long k_in = 0;                              // next window of that series
        int n_chunk = 0;
        std::vector<int64_t> coff;
        while (ser < n_seg) {
            // pieces of this chunk: [rest of series ser from window k_in] [whole series ...] [head of a long one]
            coff.clear();
            const long long s_first = seg_offsets[ser] + k_in;       // first sample the chunk reads
            coff.push_back(0);
            const int ser0 = ser;
            long long have = 0, kwin = 0;
            while (ser < n_seg) {
                const long long N = seg_offsets[ser + 1] - seg_offsets[ser];
                const long long Kser = std::max<long long>(0, N - W);
                const long long rest = N - k_in;         // samples from the piece's start to the end of the series
                if (!may_cut || have + rest <= chunk_samples || Kser - k_in <= 2 * (long long)W) {
                    have += rest; kwin += Kser - k_in;   // the whole rest of the series
                    coff.push_back(have);
                    ++ser; k_in = 0;
                    if (have >= chunk_samples) break;
                } else {
                    const long long take = std::max<long long>(chunk_samples - have - W, (long long)W);   // windows of the head piece
                    const long long kk = std::min<long long>(take, Kser - k_in);
                    have += kk + W; kwin += kk;
                    coff.push_back(have);
                    k_in += kk;
                    if (k_in >= Kser) { ++ser; k_in = 0; }
                    break;
                }
            }
            const int n_piece = (int)coff.size() - 1;
            const long long s_end = s_first + have;      // one past the last sample the chunk reads
            if (s_end > copied) {
                hipEvent_t ev = c->copy_ev[n_chunk % drl_ctx::SEG_SLOTS];
                HIP_TRY(c, hipMemcpyAsync(d_in + copied * F, series + copied * F, (size_t)(s_end - copied) * F * sizeof(float),
                                          hipMemcpyHostToDevice, c->copy_stream));
                if (explicit_flags)
                    HIP_TRY(c, hipMemcpyAsync(d_fl + copied, o->flags + copied, (size_t)(s_end - copied) * sizeof(float),
                                              hipMemcpyHostToDevice, c->copy_stream));
                HIP_TRY(c, hipEventRecord(ev, c->copy_stream));
                HIP_TRY(c, hipStreamWaitEvent(c->stream, ev, 0));
                copied = s_end;
            }
            if (kwin > 0) {
                OptDev od = od_all;                      // piece j of the chunk belongs to series ser0 + j
                if (od.in_scale) { od.in_scale += ser0; od.in_offset += ser0; }
                rc = score_impl(c, d_in + s_first * F, coff.data(), n_piece, F, W, o, d_fl ? d_fl + s_first : nullptr,
                                d_eps ? d_eps + (size_t)win_done * c->vae.latent : nullptr,
                                recon_err ? (float*)c->st_err.p + win_done : nullptr,
                                want_q ? (float*)c->st_q.p + (size_t)win_done * S * A : nullptr,
                                action ? (int32_t*)c->st_act.p + win_done : nullptr,
                                margin ? (float*)c->st_margin.p + win_done : nullptr, nullptr, false, &od);
                if (rc) return drain(rc);
                win_done += kwin;
            }
            ++n_chunk;
            chunk_samples = std::min(chunk_samples * 2, chunk_cap);
        }
        if (win_done != K)
            return drain(fail(c, DRL_ERR_STATE, "drl_score_batch: the chunk plan covered %lld of %lld windows", win_done, K));
    }
    if (recon_err) HIP_TRY(c, hipMemcpyAsync(recon_err, c->st_err.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (q) HIP_TRY(c, hipMemcpyAsync(q, c->st_q.p, (size_t)K * S * A * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (action) HIP_TRY(c, hipMemcpyAsync(action, c->st_act.p, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (margin) HIP_TRY(c, hipMemcpyAsync(margin, c->st_margin.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DRL_OK;
}

int drl_score_series(drl_ctx* c, const float* series, int64_t N, int F, int W, const drl_score_opts* o,
                     float* recon_err, float* q, int32_t* action, float* margin) {
    const int64_t off[2] = {0, N};
    return drl_score_batch(c, series, off, 1, F, W, o, recon_err, q, action, margin);
}

int drl_vae_predict(drl_ctx* c, const float* x, int64_t B, const float* eps, float* recon, float* recon_err) {
    if (!c) return DRL_ERR_ARG;
    if (!c->vae.set) return fail(c, DRL_ERR_STATE, "drl_vae_predict: drl_set_vae was not called");
    if (!x || B < 0 || (!recon && !recon_err)) return fail(c, DRL_ERR_ARG, "drl_vae_predict: bad arguments");
    if (B == 0) return DRL_OK;
    if (eps && (!c->vae.has_lv || c->vae.mu_idx < 1 || c->vae.latent > 32))
        return fail(c, DRL_ERR_UNSUPPORTED, "eps path needs a logvar layer, an encoder hidden layer and latent <= 32");
    HIP_TRY(c, hipSetDevice(c->device));
    const int D = c->vae.D;
    int rc;
    const size_t nb = (size_t)B * D * sizeof(float), eb = (size_t)B * sizeof(float);
    const size_t nz = eps ? (size_t)B * c->vae.latent * sizeof(float) : 0;
    // the reference's per-step call (one window): inputs and outputs in mapped host memory, no copies
    // (narrow windows only: a wide window is read tile by tile, one PCIe round trip each)
    const bool direct = B <= 32 && D <= 128 && 2 * up16(nb) + up16(eb) + up16(nz) <= PIN_IO_MAX;
    const float *d_x, *d_eps = nullptr;
    float *d_rec, *d_err;
    if (direct) {
        if ((rc = ensure_pin_io(c, 2 * up16(nb) + up16(eb) + up16(nz)))) return rc;
        char* base = (char*)c->pin_io;
        memcpy(base, x, nb);
        d_x = (const float*)base;
        d_rec = (float*)(base + up16(nb));
        d_err = (float*)(base + 2 * up16(nb));
        if (eps) {
            memcpy(base + 2 * up16(nb) + up16(eb), eps, nz);
            d_eps = (const float*)(base + 2 * up16(nb) + up16(eb));
        }
    } else {
        if ((rc = ensure(c, c->st_in, nb))) return rc;
        if ((rc = ensure(c, c->st_q, nb))) return rc;           // reconstruction staging
        if ((rc = ensure(c, c->st_err, eb))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->st_in.p, x, nb, hipMemcpyHostToDevice, c->stream));
        if (eps) {
            if ((rc = ensure(c, c->st_eps, nz))) return rc;
            HIP_TRY(c, hipMemcpyAsync(c->st_eps.p, eps, nz, hipMemcpyHostToDevice, c->stream));
            d_eps = (const float*)c->st_eps.p;
        }
        d_x = (const float*)c->st_in.p;
        d_rec = (float*)c->st_q.p;
        d_err = (float*)c->st_err.p;
    }
    const int64_t off[2] = {0, B};
    SegHost sh;
    if ((rc = build_segments(c, off, 1, 0, true, sh))) return rc;
    VaeArgs a;
    memset(&a, 0, sizeof a);
    fill_vae_args(c, a);
    a.x = d_x; a.win_stride = D; a.lag_elems = 0; a.D = D;
    a.eps = d_eps;
    a.seg.samp_off = sh.d_samp; a.seg.win_off = sh.d_win; a.seg.tile_off = sh.d_t32; a.seg.n_seg = 1; a.seg.W = 0;
    a.n_tiles = sh.tiles32;
    a.err = d_err;
    a.recon = recon ? d_rec : nullptr;
    prof_begin(c, 1, c->stream);
    if ((rc = launch_vae_chain(c, a, sh.tiles32, c->stream))) return rc;
    prof_end(c, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (!direct) {
        if (recon) HIP_TRY(c, hipMemcpyAsync(recon, c->st_q.p, nb, hipMemcpyDeviceToHost, c->stream));
        if (recon_err) HIP_TRY(c, hipMemcpyAsync(recon_err, c->st_err.p, eb, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (direct) {
        if (recon) memcpy(recon, d_rec, nb);
        if (recon_err) memcpy(recon_err, d_err, eb);
    }
    return DRL_OK;
}

int drl_q_predict(drl_ctx* c, const float* states, int64_t B, int W, float* q) {
    if (!c) return DRL_ERR_ARG;
    if (!c->lstm.set) return fail(c, DRL_ERR_STATE, "drl_q_predict: drl_set_lstm was not called");
    if (!states || !q || B < 0 || W < 1) return fail(c, DRL_ERR_ARG, "drl_q_predict: bad arguments");
    if (B == 0) return DRL_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const int I = c->lstm.I, A = c->lstm.A;
    int rc;
    const size_t nb = (size_t)B * W * I * sizeof(float), qb = (size_t)B * A * sizeof(float);
    const int gv_rows = 32 * c->lstm.NT, gv_ip = (I + 3) & ~3;
    const size_t gv_lds = ((size_t)W * gv_ip + (size_t)W * gv_rows) * sizeof(float);
    // One workgroup per state.  The MFMA kernels need a 25-step chain of whole tiles however few the
    // states (~140 us for the univariate net, more for wide inputs), a matrix-vector workgroup ~35 us,
    // so it wins for the per-step call (1-4 states) and for a replay batch alike.  Measured hand-over
    // (tools/predict_batch_sweep.py): the univariate net, whose MFMA kernel needs no projection pass, at
    // 4 states per CU (1,024: 144 against 175 us; 1,280: 198 against 162 us); the projection paths
    // later (I=39/H=128 at about 2,000 states, WaDi's I=124/H=64 not before 1,536) — 8 per CU there.
    const int rounds = c->gemv_rounds >= 0 ? c->gemv_rounds : (I == 2 && c->lstm.NT == 16) ? 4 : 8;
    const long long gemv_max = c->force_generic ? 4 : std::max<long long>(4, (long long)rounds * c->n_cu);
    const bool gemv = B <= gemv_max && W <= GEMV_WMAX && gv_lds <= 150 * 1024;
    // the matrix-vector kernel stages its state into LDS in one pass and writes A floats: it can work
    // on mapped host memory directly
    const bool direct = gemv && up16(nb) + qb <= PIN_IO_MAX;
    const float* d_in;
    float* d_q;
    if (direct) {
        if ((rc = ensure_pin_io(c, up16(nb) + qb))) return rc;
        memcpy(c->pin_io, states, nb);
        d_in = (const float*)c->pin_io;
        d_q = (float*)((char*)c->pin_io + up16(nb));
    } else {
        if ((rc = ensure(c, c->st_in, nb))) return rc;
        if ((rc = ensure(c, c->st_q, qb))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->st_in.p, states, nb, hipMemcpyHostToDevice, c->stream));
        d_in = (const float*)c->st_in.p;
        d_q = (float*)c->st_q.p;
    }
    const int64_t off[2] = {0, B};
    SegHost sh;
    if ((rc = build_segments(c, off, 1, 0, true, sh))) return rc;
    LstmArgs a;
    memset(&a, 0, sizeof a);
    fill_lstm_args(c, a);
    a.W = W; a.x = d_in; a.F = I; a.win_stride = (long long)W * I; a.row_stride = I; a.row0 = 0;
    a.flag_mode = -1; a.flags = nullptr;
    a.seg.samp_off = sh.d_samp; a.seg.win_off = sh.d_win; a.seg.tile_off = sh.d_t32; a.seg.n_seg = 1; a.seg.W = W;
    a.n_tiles = sh.tiles32; a.S = 1; a.q = d_q;
    // the segment table counts "samples" = windows here; the kernel multiplies samp_off (0) only
    const dim3 grid((sh.tiles32 + 3) / 4, 1);
    c->last_tile = 0;
    prof_begin(c, 0, c->stream);
    if (gemv) {
        // the per-step call of the reference's RL loop: a matrix-vector kernel, one workgroup per state
        if (gv_lds > c->gemv_lds_set) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_gemv_kernel<8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_gemv_kernel<16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            c->gemv_lds_set = 150 * 1024;
        }
        if (c->lstm.NT == 8) hipLaunchKernelGGL(lstm_gemv_kernel<8>, dim3((unsigned)B), dim3(256), gv_lds, c->stream, a);
        else hipLaunchKernelGGL(lstm_gemv_kernel<16>, dim3((unsigned)B), dim3(512), gv_lds, c->stream, a);
        c->last_variant = "gemv";
    } else if (I == 2 && c->lstm.NT == 16 && W >= 2) {
        // the register-stationary univariate kernel, reading the states row by row
        if (c->lstm.has16 && sh.tiles16 <= c->n_cu) {
            a.seg.tile_off = sh.d_t16; a.n_tiles = sh.tiles16;
            Lstm16Args fa;
            fa.l = a;
            fa.u16 = (const float*)c->lstm_w.p + c->lstm.u16_off;
            fa.wx16 = (const float*)c->lstm_w.p + c->lstm.wx16_off;
            fa.bias16 = (const float*)c->lstm_w.p + c->lstm.bias16_off;
            hipLaunchKernelGGL(lstm_h128i2_w16_kernel, dim3(sh.tiles16, 1), dim3(256), 0, c->stream, fa);
            c->last_tile = 16;
        } else if (sh.tiles32 <= c->n_cu) {
            hipLaunchKernelGGL(lstm_h128i2_kernel<1>, dim3(sh.tiles32, 1), dim3(256), 0, c->stream, a);
            c->last_tile = 32;
        } else {
            a.seg.tile_off = sh.d_t64; a.n_tiles = sh.tiles64;
            hipLaunchKernelGGL(lstm_h128i2_kernel<2>, dim3(std::min<unsigned>(sh.tiles64, c->n_cu), 1), dim3(256), 0, c->stream, a);
            c->last_tile = 64;
        }
        c->last_variant = "h128i2";
    } else if (!c->force_generic && (size_t)B * W * (c->lstm.NT == 8 ? 1024 : 2048) <= ((size_t)8 << 30)) {
        // other shapes (e.g. a replay batch of WaDi states): every row of every state is projected once
        // (the flag is just the last input column here), then the recurrence kernels of the sliding-window
        // path run with W projected rows between consecutive "windows"
        const long long n_rows = (long long)B * W;
        const int NTl = c->lstm.NT;
        if ((rc = ensure(c, c->pbuf, (size_t)n_rows * NTl * 32 * sizeof(float)))) return rc;
        XprojArgs xa;
        xa.x = a.x; xa.n_samples = n_rows; xa.F = I; xa.wx = a.wx; xa.bias = a.bias; xa.P = (float*)c->pbuf.p;
        xa.nt_total = NTl; xa.samp_off = nullptr; xa.n_seg = 0; xa.in_scale = nullptr; xa.in_offset = nullptr;
        hipLaunchKernelGGL(xproj_kernel<8>, dim3((unsigned)((n_rows + 127) / 128), NTl / 8), dim3(256), 0, c->stream, xa);
        RecArgs ra;
        ra.P = xa.P; ra.n_samples = n_rows; ra.win_rows = W;
        ra.flagw = (const float*)c->lstm_w.p + c->lstm.flagw_off;
        a.flag_mode = 0;                                    // nothing synthesised
        if (NTl == 8) {
            const size_t lds = ((size_t)32 * 8 * 64 + 8 * 32) * sizeof(float);
            if (!c->rec_attr_set) {
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec_kernel<8, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec_kernel<8, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_rec2_kernel<8>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                c->rec_attr_set = true;
            }
            ra.l = a;
            hipLaunchKernelGGL((lstm_rec_kernel<8, false>), dim3((sh.tiles32 + 3) / 4), dim3(256), lds, c->stream, ra);
            c->last_variant = "xproj+rec";
        } else if (sh.tiles32 <= c->n_cu) {
            ra.l = a;
            hipLaunchKernelGGL(lstm_h128p_kernel<1>, dim3(sh.tiles32, 1), dim3(256), 0, c->stream, ra);
            c->last_variant = "xproj+h128";
        } else {
            a.seg.tile_off = sh.d_t64; a.n_tiles = sh.tiles64;
            ra.l = a;
            hipLaunchKernelGGL(lstm_h128p_kernel<2>, dim3(std::min<unsigned>(sh.tiles64, c->n_cu), 1), dim3(256), 0, c->stream, ra);
            c->last_variant = "xproj+h128";
        }
    } else {
        if (c->lstm.NT == 8) hipLaunchKernelGGL(lstm_generic_kernel<8>, grid, dim3(256), 0, c->stream, a);
        else hipLaunchKernelGGL(lstm_generic_kernel<16>, grid, dim3(256), 0, c->stream, a);
        c->last_variant = "generic";
    }
    prof_end(c, c->stream);
    HIP_TRY(c, hipGetLastError());
    if (!direct) HIP_TRY(c, hipMemcpyAsync(q, c->st_q.p, qb, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (direct) memcpy(q, d_q, qb);
    return DRL_OK;
}

int drl_rollout_batch(drl_ctx* c, const float* series, const int64_t* seg_offsets, int n_seg, int W,
                      const float* in_scale, const float* in_offset,
                      int32_t* action, float* q, int32_t* passes_out) {
    if (!c) return DRL_ERR_ARG;
    if (!series || !seg_offsets || n_seg < 1 || W < 2 || !action)
        return fail(c, DRL_ERR_ARG, "drl_rollout_batch: bad arguments");
    if (!c->lstm.set) return fail(c, DRL_ERR_STATE, "drl_rollout_batch: drl_set_lstm was not called");
    if (c->lstm.I != 2) return fail(c, DRL_ERR_ARG, "drl_rollout_batch: the action feedback exists only for univariate states (in_dim 2)");
    if (seg_offsets[0] != 0) return fail(c, DRL_ERR_ARG, "seg_offsets[0] must be 0");
    HIP_TRY(c, hipSetDevice(c->device));
    const long long total = seg_offsets[n_seg];
    const long long K = count_windows(seg_offsets, n_seg, W);
    if (passes_out) *passes_out = 0;
    if (K == 0) return DRL_OK;
    const int A = c->lstm.A;
    int rc;
    if ((rc = ensure(c, c->st_in, (size_t)total * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->st_flags, (size_t)total * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->st_q, (size_t)K * A * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->st_act, (size_t)K * sizeof(int32_t) + 64))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->st_in.p, series, (size_t)total * sizeof(float), hipMemcpyHostToDevice, c->stream));
    // Two exact designs (measured: tools/rollout_time.py -> profiles/r2/rollout_time.txt):
    //  * fixed point over parallel scoring passes: as many passes as the busiest series has runs of action
    //    changes (+1); the first scores every window (0.15 ms for one Yahoo series, 2.5 ms for all 67), the
    //    later ones only the tiles a changed flag reaches (0.13-0.3 ms): 0.4-1.8 ms on the validation sets,
    //    9.7 ms for all 67 A1 series (33 passes; 81 ms with full passes only);
    //  * sequential kernel: one persistent workgroup per series, W dependent matrix-vector steps per cursor
    //    (19.8 us at W = 25), whatever the actions are, series in parallel one per CU: 28 ms for any batch of
    //    Yahoo-length series.
    // Auto starts with the fixed point and hands over to the sequential kernel once the passes made have cost
    // what the sequential kernel would (its time is known in advance): never more than twice the better one.
    long long longest = 0;
    for (int i = 0; i < n_seg; ++i) longest = std::max<long long>(longest, seg_offsets[i + 1] - seg_offsets[i] - W);
    const size_t seq_lds = (size_t)W * 32 * c->lstm.NT * sizeof(float);
    const bool seq_ok = seq_lds <= 150 * 1024;
    if (c->rollout_mode == 2 && !seq_ok)
        return fail(c, DRL_ERR_UNSUPPORTED, "drl_rollout_batch: the sequential kernel holds W*%d floats in LDS (W = %d is too long)", 32 * c->lstm.NT, W);
    const double seq_est_us = (double)longest * W * 0.9;
    const double pass_est_us = std::max(150.0, (double)K * W * (c->lstm.NT == 16 ? 0.00105 : 0.0006));
    const bool may_hand_over = c->rollout_mode == 0 && seq_ok;
    auto run_sequential = [&]() -> int {
        SegHost sh;
        if ((rc = build_segments(c, seg_offsets, n_seg, W, false, sh))) return rc;
        OptDev od;
        if (in_scale) {
            drl_score_opts oo;
            memset(&oo, 0, sizeof oo);
            oo.in_scale = in_scale; oo.in_offset = in_offset; oo.abi_revision = DRL_ABI_REVISION;
            if ((rc = upload_opts(c, &oo, n_seg, W, od))) return rc;
        }
        RolloutSeqArgs ra;
        memset(&ra, 0, sizeof ra);
        fill_lstm_args(c, ra.l);
        ra.l.W = W; ra.l.x = (const float*)c->st_in.p;
        ra.l.seg.samp_off = sh.d_samp; ra.l.seg.win_off = sh.d_win; ra.l.seg.tile_off = sh.d_t32; ra.l.seg.n_seg = n_seg; ra.l.seg.W = W;
        ra.l.seg.in_scale = od.in_scale; ra.l.seg.in_offset = od.in_offset;
        ra.action = (int*)c->st_act.p;
        ra.q = q ? (float*)c->st_q.p : nullptr;
        if (seq_lds > c->rollout_lds_set) {
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_seq_kernel<8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(rollout_seq_kernel<16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            c->rollout_lds_set = 150 * 1024;
        }
        if (c->lstm.NT == 8) hipLaunchKernelGGL(rollout_seq_kernel<8>, dim3(n_seg), dim3(256), seq_lds, c->stream, ra);
        else hipLaunchKernelGGL(rollout_seq_kernel<16>, dim3(n_seg), dim3(512), seq_lds, c->stream, ra);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(action, c->st_act.p, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (q) HIP_TRY(c, hipMemcpyAsync(q, c->st_q.p, (size_t)K * A * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->last_variant = "rollout_seq";
        return DRL_OK;
    };
    if (c->rollout_mode == 2) return run_sequential();     // *passes_out stays 0: no parallel passes were made
    drl_score_opts o;
    memset(&o, 0, sizeof o);
    o.struct_size = sizeof o; o.abi_revision = DRL_ABI_REVISION; o.vae_lag = 1; o.flag_mode = DRL_FLAG_EXPLICIT; o.action_rule = DRL_ACTION_ARGMAX;
    o.flags = (const float*)c->st_flags.p;
    o.in_scale = in_scale; o.in_offset = in_offset;
    SegHost sh;
    if ((rc = build_segments(c, seg_offsets, n_seg, W, false, sh))) return rc;
    // Sparse passes: a flag that changed in a pass alters only the W windows that hold its sample, so the next
    // pass scores the 16-window tiles of those windows and nothing else (the others keep their Q: exactly what a
    // full pass would write again).  The number of passes is that of the full iteration; from the second on a
    // pass costs one latency-bound launch over a few tiles instead of a scoring call of the whole batch (all 67
    // A1 series: 33 passes, 81 -> 9 ms).  Needs the register-stationary fp32 kernel family (16-window tiles).
    const bool sparse_ok = c->lstm.I == 2 && c->lstm.NT == 16 && c->lstm.has16 && c->lstm_math == 0 &&
                           W + 65 <= H128_SV && !c->rollout_dense;
    const size_t ctl_bytes = 256;                                      // ctl[0] = changed, ctl[1] = tiles listed
    if ((rc = ensure(c, c->st_margin, ctl_bytes + (sparse_ok ? (size_t)2 * sh.tiles16 * sizeof(int) : 0)))) return rc;
    SegTab seg;
    memset(&seg, 0, sizeof seg);
    seg.samp_off = sh.d_samp; seg.win_off = sh.d_win; seg.tile_off = sh.d_t16; seg.n_seg = n_seg; seg.W = W;
    float* d_flags = (float*)c->st_flags.p;
    int* d_ctl = (int*)c->st_margin.p;
    int* d_dirty = sparse_ok ? (int*)((char*)c->st_margin.p + ctl_bytes) : nullptr;
    int* d_list = sparse_ok ? d_dirty + sh.tiles16 : nullptr;
    hipLaunchKernelGGL(rollout_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_flags, seg, total);
    hipLaunchKernelGGL(rollout_init1_kernel, dim3((n_seg + 63) / 64), dim3(64), 0, c->stream, d_flags, seg);
    HIP_TRY(c, hipGetLastError());
    int passes = 0, listed = 0;
    double spent_us = 0.0;
    long long max_K = 0;
    for (int i = 0; i < n_seg; ++i) max_K = std::max<long long>(max_K, seg_offsets[i + 1] - seg_offsets[i] - W);
    for (;;) {
        // a listed tile costs a workgroup W steps of ~5 us; when the rounds of such workgroups add up to most of
        // a full pass (persistent 64-window tiles, the more efficient mapping), the full pass it is
        const double sparse_us = 30.0 + (double)((listed + c->n_cu - 1) / c->n_cu) * W * 5.0;
        const bool sparse = sparse_ok && listed > 0 && sparse_us < 0.8 * pass_est_us;
        c->tl_dev = sparse ? d_list : nullptr;
        c->tl_n = sparse ? listed : 0;
        rc = score_impl(c, (const float*)c->st_in.p, seg_offsets, n_seg, 1, W, &o, d_flags, nullptr,
                        nullptr, (float*)c->st_q.p, (int32_t*)c->st_act.p, nullptr, nullptr);
        c->tl_dev = nullptr; c->tl_n = 0;
        if (rc) return rc;
        ++passes;
        spent_us += sparse ? sparse_us : pass_est_us;
        HIP_TRY(c, hipMemsetAsync(d_ctl, 0, 2 * sizeof(int), c->stream));
        if (d_dirty) HIP_TRY(c, hipMemsetAsync(d_dirty, 0, (size_t)sh.tiles16 * sizeof(int), c->stream));
        hipLaunchKernelGGL(rollout_feedback_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, c->stream,
                           (const int*)c->st_act.p, d_flags, seg, K, d_ctl, d_dirty, d_list);
        HIP_TRY(c, hipGetLastError());
        int ctl[2] = {0, 0};
        HIP_TRY(c, hipMemcpyAsync(ctl, d_ctl, sizeof ctl, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (!ctl[0]) break;
        listed = ctl[1];
        if (may_hand_over && spent_us >= seq_est_us) {
            // auto: the passes so far have cost what stepping through the series costs — finish that way
            if (passes_out) *passes_out = passes;
            return run_sequential();
        }
        if (passes > max_K + 1)
            return fail(c, DRL_ERR_STATE, "drl_rollout_batch: no fixed point after %d passes", passes);
    }
    HIP_TRY(c, hipMemcpyAsync(action, c->st_act.p, (size_t)K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (q) HIP_TRY(c, hipMemcpyAsync(q, c->st_q.p, (size_t)K * A * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (passes_out) *passes_out = passes;
    return DRL_OK;
}

int drl_al_get_samples(drl_ctx* c, const float* series, const int64_t* seg_offsets, int n_seg, int F, int W,
                       const drl_score_opts* o, const int32_t* already_selected, int n_selected, int N,
                       int32_t* out_idx, int32_t* n_out, float* margin_out) {
    if (!c) return DRL_ERR_ARG;
    if (!series || !seg_offsets || n_seg < 1 || F < 1 || W < 1 || !out_idx || !n_out || N < 0 ||
        n_selected < 0 || (n_selected > 0 && !already_selected))
        return fail(c, DRL_ERR_ARG, "drl_al_get_samples: bad arguments");
    int rc = check_opts(c, o, F);
    if (rc) return rc;
    *n_out = 0;
    HIP_TRY(c, hipSetDevice(c->device));
    if (seg_offsets[0] != 0) return fail(c, DRL_ERR_ARG, "seg_offsets[0] must be 0");
    const long long total = seg_offsets[n_seg];
    const long long K = count_windows(seg_offsets, n_seg, W);
    if (K == 0 || N == 0) return DRL_OK;
    if (K > 0x7fffffffLL) return fail(c, DRL_ERR_ARG, "drl_al_get_samples: more than 2^31 windows");
    const int S = o->flag_mode == DRL_FLAG_BOTH ? 2 : 1;
    if ((rc = ensure(c, c->st_in, (size_t)total * F * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->st_q, (size_t)K * S * c->lstm.A * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->st_margin, (size_t)K * sizeof(float)))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->st_in.p, series, (size_t)total * F * sizeof(float), hipMemcpyHostToDevice, c->stream));
    const float* d_flags = nullptr;
    if (o->flag_mode == DRL_FLAG_EXPLICIT) {
        if ((rc = ensure(c, c->st_flags, (size_t)total * sizeof(float)))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->st_flags.p, o->flags, (size_t)total * sizeof(float), hipMemcpyHostToDevice, c->stream));
        d_flags = (const float*)c->st_flags.p;
    }
    rc = score_impl(c, (const float*)c->st_in.p, seg_offsets, n_seg, F, W, o, d_flags, nullptr, nullptr,
                    (float*)c->st_q.p, nullptr, (float*)c->st_margin.p, nullptr);
    if (rc) return rc;
    // keys = margins (already selected -> NaN, which the radix order puts last), values = window index
    for (DevBuf* b : {&c->al_keys, &c->al_keys2})
        if ((rc = ensure(c, *b, (size_t)K * sizeof(float)))) return rc;
    for (DevBuf* b : {&c->al_idx, &c->al_idx2})
        if ((rc = ensure(c, *b, (size_t)K * sizeof(int)))) return rc;
    hipLaunchKernelGGL(al_prepare_kernel, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, c->stream,
                       (const float*)c->st_margin.p, K, (float*)c->al_keys.p, (int*)c->al_idx.p);
    int n_valid_sel = 0;
    if (n_selected > 0) {
        std::vector<int32_t> uniq(already_selected, already_selected + n_selected);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        for (int32_t v : uniq) n_valid_sel += (v >= 0 && v < K);
        if ((rc = ensure(c, c->al_sel, uniq.size() * sizeof(int32_t)))) return rc;
        HIP_TRY(c, hipMemcpyAsync(c->al_sel.p, uniq.data(), uniq.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // uniq is a local
        hipLaunchKernelGGL(al_mask_kernel, dim3((unsigned)((uniq.size() + 255) / 256)), dim3(256), 0, c->stream,
                           (const int*)c->al_sel.p, (int)uniq.size(), K, (float*)c->al_keys.p);
    }
    HIP_TRY(c, hipGetLastError());
    size_t tmp_bytes = 0;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (const float*)c->al_keys.p, (float*)c->al_keys2.p,
                                                  (const int*)c->al_idx.p, (int*)c->al_idx2.p, (int)K, 0, 32, c->stream));
    if ((rc = ensure(c, c->al_tmp, tmp_bytes))) return rc;
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->al_tmp.p, tmp_bytes, (const float*)c->al_keys.p, (float*)c->al_keys2.p,
                                                  (const int*)c->al_idx.p, (int*)c->al_idx2.p, (int)K, 0, 32, c->stream));
    const long long take = std::min<long long>(N, K - n_valid_sel);
    if (take > 0)
        HIP_TRY(c, hipMemcpyAsync(out_idx, c->al_idx2.p, (size_t)take * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (margin_out)
        HIP_TRY(c, hipMemcpyAsync(margin_out, c->st_margin.p, (size_t)K * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_out = (int32_t)std::max<long long>(take, 0);
    return DRL_OK;
}

int drl_profile_enable(drl_ctx* c, int on) {
    if (!c) return DRL_ERR_ARG;
    c->prof.on = on != 0;
    return DRL_OK;
}
int drl_profile_reset(drl_ctx* c) {
    if (!c) return DRL_ERR_ARG;
    int rc = prof_collect(c);
    c->prof.ms[0] = c->prof.ms[1] = 0;
    c->prof.n[0] = c->prof.n[1] = 0;
    return rc;
}
int drl_profile_read(drl_ctx* c, int which, double* total_ms, int64_t* launches) {
    if (!c || which < 0 || which > 1) return DRL_ERR_ARG;
    int rc = prof_collect(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->prof.ms[which];
    if (launches) *launches = c->prof.n[which];
    return DRL_OK;
}
const char* drl_last_lstm_variant(drl_ctx* c) { return c ? c->last_variant : "none"; }
const char* drl_last_vae_variant(drl_ctx* c) { return c ? c->last_vae_variant : "none"; }
const char* drl_last_lstm_kernels(drl_ctx* c) { return c ? c->last_kernels.c_str() : ""; }
int drl_last_lstm_tile(drl_ctx* c) { return c ? c->last_tile : 0; }
int drl_last_call_overlapped(drl_ctx* c) { return c && c->last_overlap ? 1 : 0; }

/* ---- the path's only collective ------------------------------------------------------------------ */
namespace {
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
int rccl_load(drl_ctx* c) {
    if (g_rccl.AllGather) return DRL_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);      // the copy already in the process, if any
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(c, DRL_ERR_HIP, "RCCL is not loadable: %s", dlerror());
    c->rccl_lib = h;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy || !g_rccl.CommCount || !g_rccl.GetErrorString) {
        g_rccl = Rccl();
        return fail(c, DRL_ERR_HIP, "RCCL lacks ncclGetUniqueId/ncclCommInitRank/ncclAllGather/ncclCommCount/ncclCommDestroy");
    }
    return DRL_OK;
}
#define RCCL_TRY(c, expr)                                                                          \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess) return fail((c), DRL_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)
}  // namespace

int drl_comm_unique_id(drl_ctx* c, void* id_out) {
    if (!c || !id_out) return DRL_ERR_ARG;
    static_assert(DRL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "include/drlvae.h: DRL_COMM_ID_BYTES");
    int rc = rccl_load(c);
    if (rc) return rc;
    ncclUniqueId id;
    RCCL_TRY(c, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return DRL_OK;
}

int drl_comm_init(drl_ctx* c, int rank, int nranks, const void* id_in) {
    if (!c || !id_in || nranks < 1 || rank < 0 || rank >= nranks) return c ? fail(c, DRL_ERR_ARG, "drl_comm_init: bad rank/nranks/id") : DRL_ERR_ARG;
    if (c->comm) return fail(c, DRL_ERR_STATE, "drl_comm_init: this context already has a communicator");
    int rc = rccl_load(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    count_alloc();
    RCCL_TRY(c, g_rccl.CommInitRank(&c->comm, nranks, id, rank));
    c->comm_rank = rank; c->comm_size = nranks;
    return DRL_OK;
}

int drl_comm_size(drl_ctx* c, int* nranks_out) {
    if (!c || !nranks_out) return DRL_ERR_ARG;
    *nranks_out = 0;
    if (!c->comm) return fail(c, DRL_ERR_STATE, "drl_comm_size: drl_comm_init was not called");
    // what RCCL itself reports for the communicator it formed, not what the caller passed to drl_comm_init
    RCCL_TRY(c, g_rccl.CommCount(c->comm, nranks_out));
    return DRL_OK;
}

int drl_allgather_scores(drl_ctx* c, const void* send_dev, void* recv_dev, int64_t bytes_per_rank) {
    if (!c) return DRL_ERR_ARG;
    if (!c->comm) return fail(c, DRL_ERR_STATE, "drl_allgather_scores: drl_comm_init was not called");
    if (!send_dev || !recv_dev || bytes_per_rank < 0) return fail(c, DRL_ERR_ARG, "drl_allgather_scores: bad arguments");
    if (bytes_per_rank == 0) return DRL_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    RCCL_TRY(c, g_rccl.AllGather(send_dev, recv_dev, (size_t)bytes_per_rank, ncclInt8, c->comm, c->stream));
    return DRL_OK;
}

int drl_comm_destroy(drl_ctx* c) {
    if (!c) return DRL_ERR_ARG;
    if (!c->comm) return DRL_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    RCCL_TRY(c, g_rccl.CommDestroy(c->comm));
    c->comm = nullptr; c->comm_size = 0;
    return DRL_OK;
}

}  // extern "C"
