/*
 * libdrlvae — C-ABI of the MI355X window-scoring path (VAE reward + LSTM-DQN Q / action).
 *
 * The reference (baharehgl/Dynamic-Reward-RL-VAE) has no FFI layer: the path is reached through
 * Python call sites (vae.predict, estimator.predict, env.statefnc / env.rewardfnc, penalty_array,
 * active_learning.get_samples).  Each entry point below names the reference interface it stands
 * in for; INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every call returns 0 (DRL_OK) or a negative drl_status; the message is kept per context
 *     (drl_last_error) — nothing throws or aborts across the ABI;
 *   - a context is bound to one GPU and one HIP stream and is not thread-safe: one context per
 *     GPU / per thread;
 *   - "host" entry points take host pointers, copy in/out and return with outputs valid (the
 *     blocking semantics of sess.run / Keras predict); "_dev" entry points take device pointers
 *     on the context's GPU, enqueue on the context's stream and return without synchronising;
 *   - there is no CPU fallback: without a usable GPU drl_create fails.
 *
 * Window indexing (reference env.py:153-169, myasp-A1.py:120-134): a series of N samples and
 * window length W has K = N - W windows; window k belongs to cursor c = k + W.
 *   LSTM state of window k : rows t = 0..W-1 = samples k+1+t (includes sample c), plus a flag channel
 *   VAE window, vae_lag = 1: samples k .. k+W-1   (reference myasp-A1.py:139, excludes sample c)
 *   VAE window, vae_lag = 0: samples k+1 .. k+W   (reference myasp-wadi.py:132-135 / penalty_array[c])
 */
#ifndef DRLVAE_H
#define DRLVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct drl_ctx drl_ctx;

typedef enum drl_status {
    DRL_OK = 0,
    DRL_ERR_ARG = -1,         /* bad argument (null pointer, shape out of the supported range) */
    DRL_ERR_HIP = -2,         /* a HIP runtime call failed; message holds hipGetErrorString */
    DRL_ERR_STATE = -3,       /* weights not set / context unusable */
    DRL_ERR_UNSUPPORTED = -4  /* shape outside what the kernels implement */
} drl_status;

/* activation codes of drl_set_vae */
enum { DRL_ACT_LINEAR = 0, DRL_ACT_RELU = 1, DRL_ACT_SIGMOID = 2 };

/* how the flag channel (last LSTM input feature) is filled */
typedef enum drl_flag_mode {
    DRL_FLAG_ZEROS = 0,     /* all zero */
    DRL_FLAG_QUIRK = 1,     /* env.get_states_list(): a single 1 at row W-1-k for k < W
                               (reference myasp-A1.py:126 shifted by env.py:160-168) */
    DRL_FLAG_EXPLICIT = 2,  /* per-sample flags given by the caller: row t of window k carries
                               flags[k+1+t] (teacher-forced action history, env.py:143-147) */
    DRL_FLAG_BOTH = 3       /* two candidate states per window, flag column all 0 and all 1
                               (reference myasp-wadi.py:248-254); q is [K,2,A] */
} drl_flag_mode;

typedef enum drl_action_rule {
    DRL_ACTION_ARGMAX = 0,  /* np.argmax(Q), ties -> 0 (reference myasp-A1.py:226-227) */
    DRL_ACTION_WADI = 1     /* 1 if Q(state a1)[1] > Q(state a0)[0] (reference myasp-wadi.py:328);
                               needs DRL_FLAG_BOTH */
} drl_action_rule;

/* Revision of drl_score_opts' field MEANINGS (its size alone does not show every change): a caller sets
 * opts.abi_revision = DRL_ABI_REVISION and a library built from another revision rejects the call with
 * DRL_ERR_ARG instead of misreading the struct.
 *   1  (libdrlvae 0.1, field then named `reserved`, left 0) vae_center / vae_scale were const float*
 *   2  vae_center / vae_scale are const double* (sklearn's float64 RobustScaler parameters) */
#define DRL_ABI_REVISION 2

typedef struct drl_score_opts {
    int32_t struct_size;     /* sizeof(drl_score_opts), for forward compatibility */
    int32_t vae_lag;         /* 0 or 1, see "Window indexing" */
    int32_t flag_mode;       /* drl_flag_mode */
    int32_t action_rule;     /* drl_action_rule */
    const float* flags;      /* DRL_FLAG_EXPLICIT: one flag per sample, laid out like the series
                                (first feature stride), else NULL */
    const float* eps;        /* optional N(0,1) draws [K_total, latent] for z = mu + exp(lv/2)*eps
                                (reference myasp-A1.py:84); NULL = mean path z = mu */
    const float* in_scale;   /* optional per-series MinMax scale_ [n_seg] (reference env.py:78-79:
                                v = x*scale_ + min_, two float32 roundings); NULL = series is
                                already normalised */
    const float* in_offset;  /* per-series min_ [n_seg]; required iff in_scale */
    const double* vae_center;/* optional RobustScaler center_ [W*F], float64 as sklearn holds scale_ (and, for
                                float64 training windows, center_): reference myasp-wadi.py:96, 138.  The
                                device applies sklearn's in-place float32 transform: x - center_ rounded to
                                float32, then / scale_ rounded to float32 (see robust_apply, kernels.hip.h) */
    const double* vae_scale; /* RobustScaler scale_ [W*F], finite and non-zero; required iff vae_center */
    float vae_clip;          /* clip of the scaled VAE input (myasp-wadi.py:139); <= 0 = none */
    int32_t abi_revision;    /* DRL_ABI_REVISION of the header the caller was compiled against */
} drl_score_opts;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Creates a context on GPU `device_id` with its own stream.  Fails (DRL_ERR_HIP) when the device
 * is not a usable gfx950 GPU. */
int drl_create(int device_id, drl_ctx** out);
void drl_destroy(drl_ctx* ctx);
/* Message of the last failing call on ctx; with ctx == NULL the last drl_create failure of the
 * calling thread.  Never NULL. */
const char* drl_last_error(drl_ctx* ctx);
/* Library version string: "drlvae <version> (gfx950; abi <DRL_ABI_REVISION>; src <hash>)", where <hash> is the
 * first 16 hex digits of the sha256 over the library's sources at build time (csrc/Makefile) — a prebuilt
 * library that no longer matches the sources beside it is told apart by it (bench.py, tests/test_abi.py). */
const char* drl_version(void);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own;
 * NULL restores the context's stream. */
int drl_set_stream(drl_ctx* ctx, void* hip_stream);
/* Blocks until everything enqueued on the context's stream has finished. */
int drl_synchronize(drl_ctx* ctx);

/* ---- weights -------------------------------------------------------------------------------- */

/* Replaces load_model('vae_model.h5', custom_objects={'Sampling': Sampling}) (reference
 * myasp-A1.py:567) / build_vae (myasp-A1.py:87-101, myasp-wadi.py:101-117): a chain of Keras
 * Dense layers y = act(x.K + b), K [in,out] row-major.  Layer order: encoder hidden..., mu,
 * logvar, decoder...; dims has n_layers+1 entries for that *chain view* (dims[mu_layer+1] =
 * dims[logvar_layer+1] = latent size, and the decoder starts from the latent).  Host pointers;
 * copied and re-packed for the MFMA kernels. */
int drl_set_vae(drl_ctx* ctx, int n_layers, const int32_t* in_dims, const int32_t* out_dims,
                const int32_t* acts, int mu_layer, int logvar_layer,
                const float* const* kernels, const float* const* biases, float logvar_clip);

/* Replaces Q_Estimator_Nonlinear.__init__ + Saver.restore (reference myasp-A1.py:165-179,
 * 313-319) / QNet.build (myasp-wadi.py:212-224): TF1 LSTMCell(hidden, forget_bias) under
 * static_rnn with zero initial state, kernel [in_dim+hidden, 4*hidden] with gate column blocks
 * i|j|f|o, followed by Q = h_T.w_out + b_out.  in_dim = F + 1 (features + the action flag); a network
 * without the flag channel (in_dim = F: the legacy RLVAL / AdaptiveTauRL estimators) is scored with
 * DRL_FLAG_ZEROS.  Limits: hidden 1..128 (the reference uses 128 and 64), n_actions 2..8; beyond them
 * DRL_ERR_UNSUPPORTED / DRL_ERR_ARG. */
int drl_set_lstm(drl_ctx* ctx, int in_dim, int hidden, int n_actions, float forget_bias,
                 const float* kernel, const float* bias, const float* w_out, const float* b_out);

/* Arithmetic of the recurrent product h.U in the univariate H=128 LSTM kernel (no counterpart in
 * the reference; TensorFlow computes it in fp32):
 *   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the default;
 *   1  3-term error-compensated f16 split (h_hi.U_hi + h_hi.U_lo + h_lo.U_hi with fp32 accumulate)
 *      on the half-precision matrix pipe: same accuracy class as fp32 (tests hold it to the same
 *      1e-5 tolerance), ~5x fewer matrix cycles.  Other kernels ignore the setting. */
int drl_set_lstm_math(drl_ctx* ctx, int mode);

/* ---- the fused path ------------------------------------------------------------------------- */

/* Scores every sliding window of n_seg independent series (windows never cross a series).
 * Replaces, per window, env.rewardfnc's vae.predict + np.mean(np.square(..)) (reference
 * myasp-A1.py:139-141; batched form myasp-wadi.py:143-146), env.statefnc + estimator.predict
 * (myasp-A1.py:120-134, 197-199, 254), np.argmax / the WaDi rule (myasp-A1.py:227,
 * myasp-wadi.py:328) and the AL margin |q0-q1| (myasp-A1.py:255).
 *
 *   series      [total_samples, F] row-major, series i occupies samples seg_offsets[i] ..
 *               seg_offsets[i+1]-1 (seg_offsets has n_seg+1 entries, host memory in both variants)
 *   outputs     concatenated per series, K_i = max(0, N_i - W) windows each:
 *               recon_err [K_total], q [K_total, S, A] (S = 2 for DRL_FLAG_BOTH else 1),
 *               action [K_total] int32, margin [K_total]; any output pointer may be NULL.
 *               The margin is |q[0]-q[1]| (argmax rule) or |Q(a1)[1]-Q(a0)[0]| (WaDi rule).
 * drl_score_batch: every data pointer is host memory; synchronous.
 * drl_score_batch_dev: series / opts->flags / opts->eps / outputs are device pointers (the small
 * per-series and per-column option arrays stay host pointers); asynchronous on the stream. */
int drl_score_batch(drl_ctx* ctx, const float* series, const int64_t* seg_offsets, int n_seg,
                    int F, int W, const drl_score_opts* opts,
                    float* recon_err, float* q, int32_t* action, float* margin);
int drl_score_batch_dev(drl_ctx* ctx, const float* series, const int64_t* seg_offsets, int n_seg,
                        int F, int W, const drl_score_opts* opts,
                        float* recon_err, float* q, int32_t* action, float* margin);
/* drl_score_batch_dev for callers that cannot allocate — tight serving loops, harnesses that forbid
 * hipMalloc.  It has NO counterpart in the reference.  Same arguments, same kernels and bit-identical
 * results, but everything the call needs on the device beyond the caller's buffers (per-series tables,
 * option tables, the per-sample input projections of the multivariate LSTM paths) lives in a file-scope
 * array of 32 MiB that each GPU the library runs on holds until the process ends; the call makes no
 * hipMalloc / hipFree / hipHostMalloc, creates no events or streams and changes no function attribute
 * (drl_profile_* does not see it; drl_debug_alloc_count does not move across it — tested).
 * It is NOT capturable into a HIP graph: the tables are written on the host into one pinned block per
 * context at call time and copied from there, and the next call waits for that copy on the host.
 * Limits, rejected with DRL_ERR_ARG and a message:
 *   n_seg <= 1024;  W*F <= 8192;  total samples <= 32,512 when the LSTM is a projection path with
 *   hidden <= 64 (any F > 1, e.g. the WaDi / SMD QNet), <= 16,256 with hidden 65..128; the
 *   univariate H <= 128 network (in_dim 2, F = 1) keeps no projections and has no sample limit.
 * Every context on a GPU shares that GPU's array, which keeps the last call's contents: calls that
 * overlap in time on one GPU (two contexts, or one context on two streams) race — serialise them. */
int drl_score_batch_dev_noalloc(drl_ctx* ctx, const float* series, const int64_t* seg_offsets, int n_seg,
                                int F, int W, const drl_score_opts* opts,
                                float* recon_err, float* q, int32_t* action, float* margin);
/* One series: drl_score_batch with n_seg = 1. */
int drl_score_series(drl_ctx* ctx, const float* series, int64_t N, int F, int W,
                     const drl_score_opts* opts,
                     float* recon_err, float* q, int32_t* action, float* margin);

/* ---- call-surface compatibility (explicit batches, host pointers, synchronous) --------------- */

/* vae.predict(x) (reference myasp-A1.py:140, myasp-wadi.py:145): x [B,D] -> reconstruction
 * [B,D]; recon_err [B] (optional) = mean((recon-x)^2, axis=1); eps optional [B,latent]. */
int drl_vae_predict(drl_ctx* ctx, const float* x, int64_t B, const float* eps,
                    float* recon, float* recon_err);
/* estimator.predict(state) (reference myasp-A1.py:197-199, myasp-wadi.py:228-229):
 * states [B,W,in_dim] -> q [B,A]. */
int drl_q_predict(drl_ctx* ctx, const float* states, int64_t B, int W, float* q);

/* ---- sequential validator rollout --------------------------------------------------------------- */

/* Replaces q_learning_validator's inner loop (reference myasp-A1.py:471-496 driving env.reset /
 * env.step, env.py:82-94, 126-151) for univariate series: start from the initial state (flags
 * [0]*(W-1)+[1]), act greedily (np.argmax, ties -> 0), feed each action back as the flag of the
 * next sample.  Computed as a fixed-point iteration over full-parallel teacher-forced passes
 * (exactly the sequential result; see rollout_feedback_kernel); from the second pass on only the
 * windows that hold a sample whose flag changed are scored again.  Host pointers, synchronous.
 *   action [K_total] (required), q [K_total, A] (optional) = Q of the state the policy saw at each
 *   cursor; passes_out (optional) = number of parallel passes it took.
 * in_scale/in_offset: optional per-series MinMax parameters as in drl_score_opts. */
int drl_rollout_batch(drl_ctx* ctx, const float* series, const int64_t* seg_offsets, int n_seg, int W,
                      const float* in_scale, const float* in_offset,
                      int32_t* action, float* q, int32_t* passes_out);
/* Which of the two exact designs drl_rollout_batch runs (no counterpart in the reference):
 *   1  the fixed point over parallel scoring passes described above (passes_out = their number);
 *   2  one persistent workgroup per series that steps through the cursors, the action written straight
 *      into the next state's flag (passes_out = 0); series run in parallel, one per CU;
 *   0  (default) starts as 1 and hands over to 2 once the passes made have cost what 2 would (known in
 *      advance: 0.9 us per cursor and time step of the longest series; the passes are priced by the tiles
 *      they score): at most twice the better of the two.
 *      passes_out = the passes made before the hand-over.
 * Both give the sequential rollout's actions and the same Q, bit for bit.  Environment: DRLVAE_ROLLOUT. */
int drl_set_rollout_mode(drl_ctx* ctx, int mode);

/* ---- active-learning selection -------------------------------------------------------------------- */

/* Replaces active_learning.get_samples (reference myasp-A1.py:250-259): scores every window
 * (as drl_score_batch with the same series/opts), ranks them by the margin |q0-q1| ascending on
 * the device (stable radix sort: equal margins keep ascending window order; np.argsort's
 * quicksort leaves the order of exact ties unspecified), drops the indices in already_selected
 * and returns the first N.  Host pointers, synchronous.
 *   out_idx [N] window indices (concatenated-output numbering), *n_out = how many were written,
 *   margin_out optional [K_total] = the margins themselves. */
int drl_al_get_samples(drl_ctx* ctx, const float* series, const int64_t* seg_offsets, int n_seg, int F, int W,
                       const drl_score_opts* opts, const int32_t* already_selected, int n_selected, int N,
                       int32_t* out_idx, int32_t* n_out, float* margin_out);

/* ---- the only collective of the path ------------------------------------------------------------ */

/* The reference is single-process; windows shard across the GPUs of a node with weights
 * replicated (one process and one drl_ctx per GPU) and the per-window scores are all-gathered over
 * RCCL / xGMI (BASELINE.json north_star).  No counterpart in the reference.
 *   drl_comm_unique_id  rank 0 draws an id (DRL_COMM_ID_BYTES bytes) and hands it to the other ranks
 *                       by whatever channel the host has (torch.distributed store, MPI, a file);
 *   drl_comm_init       every rank, with that id: collective, blocks until all nranks have called;
 *   drl_allgather_scores  send_dev [bytes_per_rank] from every rank lands at
 *                       recv_dev + r*bytes_per_rank on every rank; device pointers; asynchronous on
 *                       the context's stream, i.e. ordered after the scoring call that filled send_dev;
 *   drl_comm_destroy    also done by drl_destroy.
 * RCCL is loaded on first use (dlopen), so the library loads on machines without it. */
#define DRL_COMM_ID_BYTES 128
int drl_comm_unique_id(drl_ctx* ctx, void* id_out);
int drl_comm_init(drl_ctx* ctx, int rank, int nranks, const void* id);
int drl_allgather_scores(drl_ctx* ctx, const void* send_dev, void* recv_dev, int64_t bytes_per_rank);
int drl_comm_destroy(drl_ctx* ctx);
/* The number of ranks RCCL itself reports for the context's communicator (ncclCommCount) — evidence that
 * the N-rank communicator was formed, as opposed to the nranks the caller asked drl_comm_init for. */
int drl_comm_size(drl_ctx* ctx, int* nranks_out);

/* ---- measurement ---------------------------------------------------------------------------- */

/* With profiling on, every kernel launch of the scoring path is bracketed by hipEvents on the
 * launch stream; drl_profile_read synchronises and returns accumulated milliseconds and launch
 * counts since the last drl_profile_reset.  which: 0 = LSTM kernel, 1 = VAE kernel. */
int drl_profile_enable(drl_ctx* ctx, int on);
int drl_profile_reset(drl_ctx* ctx);
int drl_profile_read(drl_ctx* ctx, int which, double* total_ms, int64_t* launches);
/* Name of the LSTM kernel variant the last call used: "h128i2" (univariate, U stationary in registers),
 * "h128i2_f16x3" (the same with drl_set_lstm_math 1), "xproj+rec" (any input width, hidden <= 64),
 * "xproj+h128" (any input width, hidden 65..128), "gemv" (drl_q_predict on few states), "generic"
 * (fallback: projections unavailable), "rollout_seq" (drl_rollout_batch's sequential kernel). */
const char* drl_last_lstm_variant(drl_ctx* ctx);
/* The LSTM kernels the last fused scoring call launched, by the names rocprofv3 prints (without "void "),
 * ';'-separated, e.g. "xproj_kernel<8>;lstm_rec2_kernel<8>;lstm_rec_kernel<8, true>": bench.py prices
 * roofline.traffic from exactly these kernels' rows of the committed PMC summary. */
const char* drl_last_lstm_kernels(drl_ctx* ctx);
/* Name of the VAE kernel the last call used: "wide16s" (wide windows staged through LDS, software-pipelined;
 * mean path and, with eps, the sampled z), "wide16" (the same layout without the pipelining: knob vae_s = 0;
 * mean path only), "wide16pp" (knob vae_pp), "chain" (narrow windows,
 * explicit batches, and calls with eps that no wide kernel takes), "chain/lds" (the same with its weights
 * copied to LDS first: calls of at most one workgroup per CU). */
const char* drl_last_vae_variant(drl_ctx* ctx);
/* Windows per workgroup tile of that launch for the "h128i2" kernels (16, 32 or 64: chosen from the
 * size of the call; results are bit-identical across the three), 0 for the other variants. */
int drl_last_lstm_tile(drl_ctx* ctx);
/* 1 when the last scoring call ran its VAE kernel on the library's second stream, concurrently with
 * the LSTM kernels (large wide-window calls; environment DRLVAE_OVERLAP = 0 never, 1 default,
 * 2 always).  Kernel durations read with drl_profile_read then overlap in time.  No counterpart in
 * the reference. */
int drl_last_call_overlapped(drl_ctx* ctx);

/* ---- test and measurement knobs (no counterpart in the reference) ----------------------------------- */

/* Sets one of the context's knobs; the environment variables DRLVAE_<NAME IN CAPITALS> give their initial
 * values and are read once, inside drl_create (no entry point calls getenv).  Names and values:
 *   force_generic 0|1      explicit-state batches always take the fallback LSTM kernel
 *   overlap 0|1|2          VAE kernel on the second stream: never | large and small calls (default) | always
 *   rec2 0|1|2             two candidate states in one wave: never | whole rounds of tiles (default) | always
 *   rollout 0|1|2          as drl_set_rollout_mode
 *   gemv_rounds n          drl_q_predict batches of up to n states per CU take the matrix-vector kernel (-1: default)
 *   vw16_pad / rec_lds_pad extra LDS bytes per workgroup of vae_wide16 / lstm_rec (occupancy experiments)
 *   pbuf_max bytes         behave as if a larger projection buffer could not be allocated (-1: off)
 *   pipe_min bytes, pipe_chunk samples, no_pipeline 0|1    the pipelined host path of drl_score_batch
 *   no_fused_epilogue 0|1  action / margin always by finalize_kernel
 *   rollout_dense 0|1      drl_rollout_batch: full passes only
 *   vae_s 0|1              wide windows: 1 default, the software-pipelined kernel (vae_wide16s_kernel) wherever one of its
 *                          workgroups fits a CU's LDS; 0: vae_wide16_kernel instead (same bits; tests, A/B)
 *   h128_split 0|1         lstm_h128i2_kernel / lstm_h128p_kernel: a last round of 64-window tiles that is at most half
 *                          full runs as 32-window tiles in a second launch (default 1; same bits, tested)
 *   vae_pp 0|1|2           wide windows: the ping-pong kernel (vae_wide16pp_kernel; same bits, measured slower):
 *                          never (default) | from a full round of tile pairs | always
 * Unknown names: DRL_ERR_ARG. */
int drl_set_knob(drl_ctx* ctx, const char* name, long long value);
/* How many times the library has asked the HIP runtime (or RCCL) for a resource since it was loaded: device or
 * pinned memory, an event, a stream, a communicator.  Process-wide.  Tests assert that it does not move across
 * drl_score_batch_dev_noalloc. */
long long drl_debug_alloc_count(void);

#ifdef __cplusplus
}
#endif
#endif /* DRLVAE_H */
