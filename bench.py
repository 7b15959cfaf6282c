#!/usr/bin/env python3
"""Throughput of the window-scoring path (VAE reward + LSTM-DQN Q / action / margin per window).

One "step" = one pass of the hot path over one batch of series resident in HBM
(drl_score_batch_dev).  Workloads:
  N = 1 default        C2: the 67 Yahoo A1 series (committed fixture, normalised by the reference's env.py)
                       with the reference's trained checkpoint weights — BASELINE.json configs[1];
  N > 1 default        C5: a synthetic univariate stream, W = 50, weak-scaled (every rank its own stream of
                       --windows windows, default 4e6 per GPU per step; north_star: 1.25e8);
  --workload c4 --shard   one 172,801-sample WaDi-shaped stream strong-scaled over the ranks
                       (drlvae.dist.shard_stream: contiguous window ranges with a W-1 sample halo).
With N > 1 the per-window scores are all-gathered over RCCL (one packed block per rank per step,
drlvae.dist.ScorePack), the only collective of the path.

At N = 1 the same JSON line also carries (never in `value`):
  other_configs        every other BASELINE config on this GPU — C5 (4e6 windows per step, >= 1 s of sustained
                       load), C3, C4, C1 — each with its rate, ms per step, LSTM kernel, executed roofline
                       fraction and, where the call overlaps its kernels, each kernel's own duration;
  host_path            drl_score_batch (pageable numpy in, numpy out: the reference's drop-in call) on C2 and C4;
  library              drl_version (with the hash of the sources it was built from) against the tree's hash.
They ride on the flag-less run only (an explicit --workload is one workload); --no-others also drops
roofline.kernels_alone (the PMC passes use it: one launch population per kernel).

`python bench.py --gpus N` from a bare shell starts the N ranks itself (a child
`python -m torch.distributed.run`, before anything touches the GPU); under a launcher (WORLD_SIZE set)
it is one rank.  Prints ONE JSON line on rank 0 (see README/DESIGN.md for the fields).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: 256 CU x 2.4 GHz x 256 flop/clk


def glorot(rng, a, b):
    lim = np.sqrt(6.0 / (a + b))
    return rng.uniform(-lim, lim, (a, b)).astype(np.float32)


def make_vae(rng, chain, mu_pos):
    """chain e.g. [25,64,64,64,10,64,25]; layer mu_pos maps to the latent."""
    ks, bs, acts = [], [], []
    for i in range(len(chain) - 1):
        ks.append(glorot(rng, chain[i], chain[i + 1]))
        bs.append(np.zeros(chain[i + 1], np.float32))
        acts.append(1)
        if i == mu_pos:
            acts[-1] = 0
            ks.append(glorot(rng, chain[i], chain[i + 1]))
            bs.append(np.zeros(chain[i + 1], np.float32))
            acts.append(0)
    acts[-1] = 2
    return {"kernels": ks, "biases": bs, "acts": acts, "mu_layer": mu_pos, "logvar_layer": mu_pos + 1,
            "logvar_clip": 10.0}


def make_lstm(rng, I, H, A=2):
    return {"kernel": glorot(rng, I + H, 4 * H), "bias": np.zeros(4 * H, np.float32),
            "w_out": glorot(rng, H, A), "b_out": np.zeros(A, np.float32), "forget_bias": 1.0}


def a1_fixture():
    """The 67 Yahoo A1 series as the reference's env.py normalises them, and the reference's trained
    checkpoint (tests/golden/make_golden.py wrote both; they travel with the repository)."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "a1_series.npz"))
    w = np.load(os.path.join(ROOT, "tests", "golden", "ckpt_a1.npz"))
    lstm = {k[5:]: w[k] for k in w.files if k.startswith("lstm_")}
    lstm["forget_bias"] = 1.0
    n = len(w["vae_acts"])
    vae = {"kernels": [w[f"vae_k{i}"] for i in range(n)], "biases": [w[f"vae_b{i}"] for i in range(n)],
           "acts": [int(a) for a in w["vae_acts"]], "mu_layer": int(w["vae_mu_layer"]),
           "logvar_layer": int(w["vae_logvar_layer"]), "logvar_clip": float(w["vae_logvar_clip"])}
    return z["values"].astype(np.float32), z["offsets"].astype(np.int64), vae, lstm


def workload(name, rank, windows, hidden=0, shard=None):
    """Shapes of SURVEY.md §8(d).  Returns dict(series [N,F] float32, offsets, W, F, vae, lstm, opts).
    shard = (rank, world): strong scaling, this rank's halo'd slice of the one stream (c4)."""
    rng_w = np.random.default_rng(1234)
    data = "synthetic"
    lstm = None
    if name in ("c1", "c2"):
        W, F, H = 25, 1, 128
        values, off, vae, lstm = a1_fixture()
        n_ser = 67 if name == "c2" else 1
        lens = np.diff(off)[:n_ser]
        series = values[:off[n_ser], None]
        data = ("real: Yahoo A1 series normalised by the reference's env.py + the reference's trained checkpoint "
                "(committed fixtures); every rank scores the same series")
        desc = ("C2: all 67 Yahoo A1 series in one ragged batch (93,191 windows), W=25, univariate, "
                "VAE 25-64-64-64-10-64-25 + LSTM I2/H128" if name == "c2" else
                "C1: Yahoo A1 real_1 (1,395 windows), W=25")
        kw = dict(vae_lag=1, flag_mode="quirk", action_rule="argmax")
    elif name == "c5":
        W, F, H = 50, 1, 128
        n = int(windows) + W
        rng = np.random.default_rng(50 + rank)
        series = rng.random((n, 1), dtype=np.float32)
        lens = np.array([n])
        vae = make_vae(rng_w, [50, 64, 64, 64, 10, 64, 50], 3)
        desc = f"C5: one synthetic univariate stream of {int(windows):,} windows per GPU, W=50, LSTM I2/H128"
        kw = dict(vae_lag=1, flag_mode="zeros", action_rule="argmax")
    elif name in ("c3", "c4"):
        W = 25
        F, n, seed = (38, 28479, 38) if name == "c3" else (123, int(windows) + W if windows else 172801, 123)
        H = hidden or 64                 # SURVEY 8d lists H=64 (WaDi QNet) and H=128 (the reference's default)
        rng = np.random.default_rng(seed + (0 if shard else rank))
        if name == "c3":
            series = rng.random((n, F), dtype=np.float32)
        else:
            series = (rng.standard_normal((n, F), dtype=np.float32) *
                      rng.uniform(0.1, 100, F).astype(np.float32))
        lens = np.array([n])
        vae = make_vae(rng_w, [W * F, 128, 10, 128, W * F], 1)
        samp = rng.choice(np.arange(W, n), 200, replace=False)
        wins = np.stack([series[i - W + 1:i + 1].reshape(-1) for i in samp]).astype(np.float64)
        center = np.median(wins, axis=0)
        q = np.percentile(wins, [25, 75], axis=0)
        scale = q[1] - q[0]
        scale[scale == 0] = 1.0
        desc = (f"{name.upper()}: synthetic {F}-sensor stream, N={n:,}, W=25, VAE {W*F}-128-10-128-{W*F}, "
                f"LSTM I{F+1}/H{H}, two candidate states per window")
        kw = dict(vae_lag=0, flag_mode="both", action_rule="wadi", vae_center=center, vae_scale=scale,
                  vae_clip=10.0)
        if shard:
            from drlvae.dist import shard_stream
            r, world = shard
            s_lo, s_hi, k_lo, k_hi = shard_stream(n, W, r, world, 0)
            series = np.ascontiguousarray(series[s_lo:s_hi])
            lens = np.array([s_hi - s_lo])
            desc += (f"; ONE stream sharded over {world} GPUs (strong scaling): rank {r} scores windows "
                     f"[{k_lo:,}, {k_hi:,}) from samples [{s_lo:,}, {s_hi:,}) (W-1 sample halo)")
    else:
        raise SystemExit(f"unknown workload {name}")
    if lstm is None:
        lstm = make_lstm(rng_w, F + 1, H)
    if name == "c4":
        lstm["kernel"][:F] *= 0.01
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    S = 2 if kw["flag_mode"] == "both" else 1
    chain = [k.shape[0] for i, k in enumerate(vae["kernels"]) if i != vae["logvar_layer"]] + [W * F]
    vae_flops = 2 * sum(chain[i] * chain[i + 1] for i in range(len(chain) - 1))
    lstm_flops = S * (2 * W * (F + 1 + H) * 4 * H + 2 * H * 2)
    return dict(name=name, desc=desc, series=series, offsets=offsets, W=W, F=F, H=H, S=S, vae=vae, lstm=lstm,
                kw=kw, vae_flops=vae_flops, lstm_flops=lstm_flops, data=data,
                K=int(np.maximum(lens - W, 0).sum()))


def cpu_baseline(wl):
    """The oracle (fp32 CPU restatement of the reference's path, batched per series) timed on the host
    cores of this box on a bounded sample of the same workload: the numpy restatement (oracle/ref_path.py,
    BLAS threads) is the headline leg, the torch-CPU one (oracle/ref_path_torch.py) rides beside it
    (SURVEY.md 8d names both), and ~1 s of the reference's own batch-1 call pattern: ~8 s in all."""
    from oracle import ref_path as orc
    from oracle import ref_path_torch as orct
    import torch
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count()
    W, F = wl["W"], wl["F"]
    x, off = wl["series"], wl["offsets"]
    uni_mode = "quirk" if wl["kw"]["flag_mode"] == "quirk" else "zeros"

    def leg(score_uni, score_multi, budget):
        done, passes, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < budget:
            for i in range(len(off) - 1):
                s = x[off[i]:off[i + 1]]
                if F == 1:
                    n = min(len(s), 20000 + W)
                    score_uni(s[:n, 0])
                else:
                    n = min(len(s), 600 + W)
                    score_multi(s[:n])
                done += max(0, n - W)
                if time.perf_counter() - t0 > budget:
                    break
            passes += 1
        return done, passes, time.perf_counter() - t0

    done, passes, dt = leg(lambda v: orc.score_series_univariate(v, W, wl["vae"], wl["lstm"], uni_mode),
                           lambda m: orc.score_series_multivariate(m, W, wl["vae"], wl["lstm"], wl["kw"]["vae_center"],
                                                                   wl["kw"]["vae_scale"], 10.0), 4.0)
    tthreads = torch.get_num_threads()
    tdone, tpasses, tdt = leg(lambda v: orct.score_series_univariate(v, W, wl["vae"], wl["lstm"], uni_mode),
                              lambda m: orct.score_series_multivariate(m, W, wl["vae"], wl["lstm"], wl["kw"]["vae_center"],
                                                                       wl["kw"]["vae_scale"], 10.0), 3.0)
    # the reference's own call pattern (one vae.predict + one estimator.predict per window, batch 1:
    # SURVEY.md 3.1), ~1 s of it, reported beside the batched figure
    b1 = None
    if F == 1:
        v = x[off[0]:off[1], 0]
        vae, lstm = wl["vae"], wl["lstm"]
        n1, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 1.0 and n1 + W + 1 < len(v):
            xw = v[n1:n1 + W][None, :]
            rec = orc.vae_forward(xw, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"], vae["logvar_layer"], 10.0)
            orc.recon_error(xw, rec)
            st = np.zeros((1, W, 2), np.float32)
            st[0, :, 0] = v[n1 + 1:n1 + 1 + W]
            orc.greedy_action(orc.lstm_q(st, lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], 1.0))
            n1 += 1
        b1 = {"value": n1 / (time.perf_counter() - t1), "unit": "windows/s",
              "sample": f"{n1} windows, one batch-1 VAE call + one batch-1 LSTM call each (the reference's call pattern)"}
    return {"value": done / dt, "unit": "windows/s", "cores": int(cores), "kind": "port", "call_pattern_batch1": b1,
            "torch_cpu_fp32": {"value": tdone / tdt, "unit": "windows/s", "cores": int(tthreads),
                               "sample": f"{tdone:,} windows ({tpasses} pass(es)) in {tdt:.1f} s; oracle/ref_path_torch.py, "
                                         f"torch {torch.__version__} CPU, {tthreads} threads"},
            "sample": f"{done:,} windows ({passes} pass(es) over the head of the same workload) in {dt:.1f} s; "
                      f"oracle/ref_path.py = numpy fp32 restatement, whole-series batches through BLAS "
                      f"(not the reference's TF path, which cannot run here; its batch-1 call pattern is "
                      f"~100x slower, BASELINE.md)"}


def pmc_traffic(name, kernels):
    """HBM-side bytes per scoring call of the LSTM kernels that ran (`kernels`: their rocprofv3 names, from
    drl_last_lstm_kernels), each kernel's per-launch average in the committed PMC summary of the same workload
    (profiles/rNN/<workload>_pmc_summary.json, newest round; made from `bench.py --no-others` runs, so every
    kernel has ONE launch population).  None when a kernel that ran has no row there."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"{name}_pmc_summary.json"))):
        best = f
    if not best or not kernels:
        return None
    try:
        rows = {k.replace("void ", "").strip(): v for k, v in json.load(open(best))["kernels"].items()}
        return float(sum(rows[k]["hbm_bytes_per_launch"] for k in kernels))
    except (KeyError, ValueError, OSError):
        return None


def executed_flop(wl, variant):
    """Flop one LSTM launch really executes: the xproj paths project each SAMPLE once, not each window's W rows."""
    H, F, W, S, K = wl["H"], wl["F"], wl["W"], wl["S"], wl["K"]
    if variant.startswith("xproj"):
        return int(wl["series"].shape[0]) * 2 * F * 4 * H + K * S * (W * 2 * H * 4 * H + 2 * H * 2)
    return wl["lstm_flops"] * K


def kernels_alone(wl, local, stream, series, out, steps):
    """Each kernel's own duration: the same call on a second context with the VAE kernel kept on the LSTM
    kernels' stream (knob overlap = 0), a few untimed steps."""
    from drlvae.scorer import WindowScorer
    sc1 = WindowScorer(local)
    try:
        sc1.set_knob("overlap", 0)
        sc1.set_vae(wl["vae"])
        sc1.set_lstm(wl["lstm"])
        sc1.set_stream(stream.cuda_stream)
        for _ in range(2):
            sc1.score_dev(series, wl["W"], wl["offsets"], out, **wl["kw"])
        sc1.synchronize()
        sc1.profile(True)
        sc1.profile_reset()
        for _ in range(steps):
            sc1.score_dev(series, wl["W"], wl["offsets"], out, **wl["kw"])
        sc1.synchronize()
        (l_ms, l_n), (v_ms, v_n) = sc1.profile_read(0), sc1.profile_read(1)
        l_avg = l_ms / max(l_n, 1)
        return {"lstm_kernels_avg_ms": l_avg, "vae_kernel_avg_ms": v_ms / max(v_n, 1), "steps": steps,
                "lstm_kernels": sc1.lstm_kernels,
                "lstm_frac_alone": executed_flop(wl, sc1.lstm_variant) / (l_avg / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS if l_n else 0.0,
                "what": "the same call with the overlap off (one stream): each kernel's own duration"}
    finally:
        sc1.close()


def side_config(name, steps, warmup, local, windows=0):
    """One of the other BASELINE configs on this GPU, its own context and stream, inputs resident in HBM:
    `steps` timed steps after `warmup`, HIP events around the kernels, then the kernels alone."""
    import torch
    from drlvae.scorer import WindowScorer
    dev = torch.device("cuda", local)
    wl = workload(name, 0, windows)
    K, S = wl["K"], wl["S"]
    stream = torch.cuda.Stream(device=dev)
    sc = WindowScorer(local)
    try:
        sc.set_vae(wl["vae"])
        sc.set_lstm(wl["lstm"])
        sc.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            series = torch.from_numpy(wl["series"]).to(dev)
            out = {"err": torch.empty(K, dtype=torch.float32, device=dev),
                   "q": torch.empty((K, S, 2), dtype=torch.float32, device=dev),
                   "action": torch.empty(K, dtype=torch.int32, device=dev),
                   "margin": torch.empty(K, dtype=torch.float32, device=dev)}
        stream.synchronize()
        for _ in range(warmup):
            sc.score_dev(series, wl["W"], wl["offsets"], out, **wl["kw"])
        sc.synchronize()
        torch.cuda.synchronize()
        sc.profile(True)
        sc.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            sc.score_dev(series, wl["W"], wl["offsets"], out, **wl["kw"])
        sc.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        (l_ms, l_n), (v_ms, v_n) = sc.profile_read(0), sc.profile_read(1)
        sc.profile(False)
        variant, overlapped = sc.lstm_variant, sc.overlapped
        l_avg = l_ms / max(l_n, 1)
        ex = executed_flop(wl, variant)
        res = {"workload": wl["desc"], "value": K * steps / dt, "unit": "windows/s", "steps": steps, "warmup": warmup,
               "ms_per_step": dt / steps * 1e3, "timed_region_s": dt, "windows_per_step": K,
               "flop_per_window": wl["vae_flops"] + wl["lstm_flops"], "lstm_kernel": variant, "lstm_kernels": sc.lstm_kernels,
               "vae_kernel": sc.vae_variant, "lstm_tile": sc.lstm_tile,
               "lstm_kernels_avg_ms": l_avg, "vae_kernel_avg_ms": v_ms / max(v_n, 1),
               "frac": ex / (l_avg / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS if l_n else 0.0,
               "frac_is": "flop the LSTM launch executes / its HIP-event time / 157.3 TFLOP/s",
               "vae_and_lstm_kernels_overlap": overlapped}
        if variant.startswith("xproj"):
            res["algorithmic_frac"] = wl["lstm_flops"] * K / (l_avg / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS if l_n else 0.0
        if overlapped:
            al = kernels_alone(wl, local, stream, series, out, max(2, min(steps, 3)))
            res["frac_alone"] = al["lstm_frac_alone"]
            res["lstm_kernels_ms_alone"] = al["lstm_kernels_avg_ms"]
            res["vae_kernel_ms_alone"] = al["vae_kernel_avg_ms"]
        else:
            res["frac_alone"] = res["frac"]
            res["lstm_kernels_ms_alone"] = l_avg
            res["vae_kernel_ms_alone"] = res["vae_kernel_avg_ms"]
        # the VAE kernel against the same peak (wide windows only: there it is a real GEMM)
        if wl["F"] > 1 and res["vae_kernel_ms_alone"] > 0:
            res["vae_frac_alone"] = wl["vae_flops"] * K / (res["vae_kernel_ms_alone"] / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS
        return res
    finally:
        sc.close()
        torch.cuda.synchronize()


def host_path(name, calls, local):
    """The drop-in call of a numpy caller (drlvae.compat.penalty_array, WindowScorer.score): drl_score_batch
    with pageable host buffers in and out, synchronous — PCIe included.  Never `value`."""
    from drlvae.scorer import WindowScorer
    wl = workload(name, 0, 0)
    sc = WindowScorer(local)
    try:
        sc.set_vae(wl["vae"])
        sc.set_lstm(wl["lstm"])
        x, off = wl["series"], wl["offsets"]
        for _ in range(2):
            sc.score(x, wl["W"], off, **wl["kw"])
        t0 = time.perf_counter()
        for _ in range(calls):
            sc.score(x, wl["W"], off, **wl["kw"])
        dt = time.perf_counter() - t0
        return {"workload": wl["desc"], "value": wl["K"] * calls / dt, "unit": "windows/s", "calls": calls,
                "ms_per_call": dt / calls * 1e3, "bytes_in": int(x.nbytes), "bytes_out": int(wl["K"] * (12 + 8 * wl["S"])),
                "entry_point": "drl_score_batch: pageable numpy series in, err/q/action/margin numpy out, synchronous"}
    finally:
        sc.close()


def library_info(sc):
    """drl_version of the loaded library (it carries the hash of the sources it was compiled from) against the
    hash of the sources in this tree: a stale prebuilt .so shows as `stale: true`."""
    import re
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from drlvae import _abi
    v = sc.version
    m = re.search(r"src ([0-9a-f]{16})", v)
    try:
        from src_hash import src_hash
        tree = src_hash()
    except Exception:
        tree = None
    return {"version": v, "path": os.path.relpath(_abi.LIB_PATH, ROOT), "src_hash_compiled_in": m.group(1) if m else None,
            "src_hash_of_tree": tree, "stale": (m.group(1) != tree) if (m and tree) else None,
            "build_mode": "prebuilt in-tree by __graft_entry__.build() (csrc/Makefile, hipcc --offload-arch=gfx950); "
                          "nothing is compiled at run time"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as a child torchrun.  This process has
    touched no GPU (torch is not even imported yet) and never does; it passes the child's output through
    and exits with its code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=["c1", "c2", "c3", "c4", "c5"],
                    help="default: c2 on one GPU, c5 (weak-scaled) on several")
    ap.add_argument("--shard", action="store_true",
                    help="strong scaling: ONE stream (c4) sharded over the ranks instead of one stream per rank")
    ap.add_argument("--windows", type=float, default=0, help="windows per GPU for c5 (default 4e6) / c4")
    ap.add_argument("--lstm-math", default="fp32", choices=["fp32", "f16x3"],
                    help="recurrent product of the univariate H=128 LSTM: exact fp32 MFMA or 3-term f16 split")
    ap.add_argument("--hidden", type=int, default=0, help="LSTM hidden width for c3/c4 (64 default, 128)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allgather", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the f16x3 side measurement")
    ap.add_argument("--no-others", action="store_true",
                    help="N = 1: skip other_configs / host_path / kernels_alone (profiling passes: one launch population per kernel)")
    ap.add_argument("--collective", default="torch", choices=["torch", "abi"],
                    help="the all-gather of the scores: torch.distributed (RCCL) or the library's own RCCL "
                         "entry point drl_allgather_scores on the scoring stream")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)                            # does not return

    import torch
    import torch.distributed as dist
    from drlvae.dist import ScorePack

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    default_workload = args.workload is None           # the driver's flag-less run: it also carries the other configs
    if args.workload is None:
        args.workload = "c2" if world == 1 else "c5"
    if args.shard and args.workload != "c4":
        raise SystemExit("--shard is the strong-scaling form of --workload c4")
    # rehearsal of the N>1 path on a one-GPU box: DRLVAE_BENCH_REHEARSE=1 puts every rank on cuda:0 and
    # swaps RCCL (which refuses two ranks on one device) for gloo.  Never a measurement.
    rehearse = os.environ.get("DRLVAE_BENCH_REHEARSE") == "1"
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if rehearse:
        local = 0
    elif ndev >= local_world:
        pass                                              # one visible device per local rank: rank i on device i
    elif ndev == 1 and world > 1:
        local = 0                                         # a launcher that masks the devices per rank; checked below
    elif ndev < local_world:
        raise SystemExit(f"bench.py: {local_world} ranks on this node but {ndev} visible GPU(s): refusing to put two "
                         f"ranks on one device (DRLVAE_BENCH_REHEARSE=1 rehearses the N>1 path on one GPU)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if rehearse else dev  # gloo has no GPU all_gather: the rehearsal stages through the host
    rccl_ranks, devices = 1, None
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        # which physical device each rank sits on: N ranks must be on N different GPUs (a per-rank device mask
        # that maps two ranks to one GPU would otherwise go unnoticed)
        pr = torch.cuda.get_device_properties(local)
        ident = str(getattr(pr, "uuid", "")) or str(getattr(pr, "pci_bus_id", ""))
        ident = f"{socket.gethostname()}:{ident}:{getattr(pr, 'pci_bus_id', '')}:{getattr(pr, 'pci_device_id', '')}"
        devices = [None] * world
        dist.all_gather_object(devices, ident)
        if not rehearse and len(set(devices)) != world:
            raise SystemExit(f"bench.py: {world} ranks share {len(set(devices))} device(s) {sorted(set(devices))}: "
                             f"every rank needs a GPU of its own")
        # ranks as the collective library itself counts them: a device-side sum of ones over the group
        # (backend "nccl" = RCCL; in the rehearsal gloo on the host)
        ones = torch.ones(1, dtype=torch.float32, device=cdev)
        dist.all_reduce(ones)
        rccl_ranks = int(round(float(ones.item())))
        if rccl_ranks != world:
            raise SystemExit(f"bench.py: the all-reduce over the process group counted {rccl_ranks} ranks, expected {world}")

    if args.workload == "c5" and not args.windows:
        args.windows = 4e6
    wl = workload(args.workload, rank, args.windows, args.hidden, (rank, world) if args.shard else None)
    from drlvae.scorer import WindowScorer
    sc = WindowScorer(local)
    sc.set_vae(wl["vae"])
    sc.set_lstm(wl["lstm"])
    sc.set_lstm_math(args.lstm_math)
    stream = torch.cuda.Stream(device=dev)
    sc.set_stream(stream.cuda_stream)

    K, S = wl["K"], wl["S"]
    gather = world > 1 and not args.no_allgather
    # windows per rank: equal under weak scaling, within one of each other when one stream is sharded
    if world > 1:
        ks = [None] * world
        dist.all_gather_object(ks, K)
    else:
        ks = [K]
    K_cap, K_total = max(ks), sum(ks)
    with torch.cuda.stream(stream):
        series = torch.from_numpy(wl["series"]).to(dev)
        # one ScorePack per rank and step: [err | q | action | margin] as 4-byte words, so that the whole
        # result travels in ONE all-gather.  Two sets of buffers: the all-gather of step i runs (on RCCL's
        # stream) under the kernels of step i+1.
        nbuf = 2 if gather else 1
        packs = [ScorePack(K_cap, S, 2, dev) for _ in range(nbuf)]
        outs = [p.views(K) for p in packs]
        for o in outs:
            if S == 1:
                o["q"] = o["q"].view(K, 1, 2)
        gathered = [torch.empty(world * packs[0].words, dtype=torch.float32, device=cdev) for _ in range(nbuf)] if gather else None
    stream.synchronize()

    use_abi = gather and args.collective == "abi" and not rehearse
    abi_ranks = None
    if use_abi:
        ids = [sc.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        sc.comm_init(rank, world, ids[0])
        abi_ranks = sc.comm_size()                        # ncclCommCount of the library's own communicator
        if abi_ranks != world:
            raise SystemExit(f"bench.py: drl_comm_size reports {abi_ranks} ranks, expected {world}")

    pending = [None] * nbuf
    counter = [0]

    def step():
        b = counter[0] % nbuf
        counter[0] += 1
        if pending[b] is not None:                       # its buffers are rewritten below
            with torch.cuda.stream(stream):
                pending[b].wait()
            pending[b] = None
        sc.score_dev(series, wl["W"], wl["offsets"], outs[b], **wl["kw"])
        if use_abi:
            sc.allgather(packs[b].buf, gathered[b])      # RCCL on the scoring stream, no torch in between
        elif gather:
            with torch.cuda.stream(stream):
                if rehearse:
                    pending[b] = dist.all_gather_into_tensor(gathered[b], packs[b].buf.cpu(), async_op=True)
                else:
                    pending[b] = packs[b].all_gather(gathered[b], async_op=True)

    def fence():
        with torch.cuda.stream(stream):
            for b in range(nbuf):
                if pending[b] is not None:
                    pending[b].wait()
                    pending[b] = None
        stream.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, profile=True):
        fence()
        if profile:
            sc.profile(True)
            sc.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        res = (sc.profile_read(0), sc.profile_read(1)) if profile else None
        if profile:
            sc.profile(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, res

    for _ in range(args.warmup):
        step()
    dt, ((lstm_ms, lstm_n), (vae_ms, vae_n)) = timed(args.steps)
    variant = sc.lstm_variant  # of the headline run (the alt run below changes it)
    lstm_kernel_names = sc.lstm_kernels
    vae_variant = sc.vae_variant
    overlapped = sc.overlapped

    # When the VAE kernel ran on the second stream beside the LSTM kernels, the per-kernel durations above share
    # one time span.  A few untimed steps on a second context with the overlap off give each kernel's own duration
    # (reported beside the roofline line, never as `value`).
    alone = None
    if overlapped and rank == 0 and not args.no_others:
        alone = kernels_alone(wl, local, stream, series, outs[0], max(2, min(args.steps, 5)))
        torch.cuda.synchronize()

    # the rate of ONE GPU on this same workload with the collective off (rank 0's own kernels, same steps):
    # what a weak-scaling efficiency of this line should be read against (the N = 1 default is another workload)
    single = None
    if gather:
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sc.score_dev(series, wl["W"], wl["offsets"], outs[0], **wl["kw"])
        fence()
        single = {"value": K * args.steps / (time.perf_counter() - t0), "unit": "windows/s",
                  "what": "rank 0 alone on its share of this workload, collective off, same step count"}

    # the same workload once more with the other arithmetic of the H=128 recurrent product, reported
    # beside the headline (never as `value`): the 3-term compensated f16 split
    alt = None
    if variant.startswith("h128i2") and args.lstm_math == "fp32" and not args.no_alt:
        sc.set_lstm_math("f16x3")
        for _ in range(max(1, args.warmup)):
            step()
        dt2, ((ms2, n2), _) = timed(args.steps)
        alt = {"lstm_math": "f16x3 (h.U as h_hi.U_hi + h_hi.U_lo + h_lo.U_hi on the f16 matrix pipe, fp32 accumulate; "
                            "same 1e-5 parity tests as fp32, measured ~3e-7 from the fp64 oracle)",
               "value": K_total * args.steps / dt2, "unit": "windows/s", "ms_per_step": dt2 / args.steps * 1e3,
               "lstm_kernel": sc.lstm_variant, "lstm_kernel_avg_ms": ms2 / max(n2, 1)}
        sc.set_lstm_math("fp32")

    if rank == 0:
        total_windows = K_total * args.steps
        lstm_avg_s = lstm_ms / max(lstm_n, 1) / 1e3
        algorithmic = wl["lstm_flops"] * K
        # what the launch really executes: the xproj paths project each SAMPLE once, not each window's W rows
        executed = executed_flop(wl, variant)
        achieved = executed / lstm_avg_s / 1e12 if lstm_n else 0.0
        res = {
            "metric": "windows scored/sec (VAE reward + LSTM Q)",
            "value": total_windows / dt,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.shard else "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.lstm_math == "fp32" else "f16x3-split MFMA, f32 accumulate",
            "data": wl["data"],
            "config": {"workload": wl["desc"], "windows_per_gpu_per_step": K, "windows_per_rank": ks,
                       "windows_per_step_all_gpus": K_total, "ranks": world,
                       "rccl_ranks": rccl_ranks,
                       "rccl_ranks_is": ("a device-side all_reduce of ones over the nccl (= RCCL) process group"
                                         if world > 1 and not rehearse else
                                         "REHEARSAL: an all_reduce of ones over gloo on the host" if world > 1 else "single process"),
                       "abi_comm_ranks": abi_ranks, "devices": devices,
                       "window": wl["W"], "features": wl["F"], "hidden": wl["H"], "candidate_states": S,
                       "flop_per_window": wl["vae_flops"] + wl["lstm_flops"],
                       "collective": ("none" if not gather else "gloo all_gather (REHEARSAL, not a measurement)"
                                      if rehearse else ("one rccl all_gather (drl_allgather_scores) of the packed err/q/action/margin block per step"
                                                       if use_abi else "one rccl all_gather of the packed err/q/action/margin block per step, overlapped with the next step's kernels")),
                       "lstm_kernel": variant, "lstm_kernels": lstm_kernel_names, "vae_kernel": vae_variant,
                       "vae_and_lstm_kernels_overlap": overlapped},
            "roofline": {"bound": "mfma", "kernel": "lstm_" + variant,
                         "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(wl["name"], lstm_kernel_names),
                         "traffic_unit": "HBM-side bytes per scoring call of exactly config.lstm_kernels (rocprofv3 FETCH_SIZE x2 "
                                         "gfx950 correction + WRITE_SIZE, per-launch averages of the committed PMC summary)",
                         "algorithmic_bytes_per_launch": (4 * wl["F"] + 4 + 4 * 2 * S + 4) * K,
                         "avg_launch_ms": lstm_avg_s * 1e3, "launches": lstm_n,
                         "flop_per_launch": executed,
                         "vae_kernel_avg_ms": vae_ms / max(vae_n, 1)},
        }
        if variant.startswith("xproj"):
            # `frac` prices the flop the launch executes.  The reference re-projects the W input rows of every
            # window (SURVEY 8d's per-window figure); for the record, that algorithmic count over the same time:
            res["roofline"]["algorithmic_flop_per_launch"] = algorithmic
            res["roofline"]["algorithmic_frac"] = algorithmic / lstm_avg_s / 1e12 / PEAK_F32_MFMA_TFLOPS if lstm_n else 0.0
        if overlapped:
            res["roofline"]["note"] = ("the VAE kernel runs on a second stream beside the LSTM kernels: avg_launch_ms and "
                                       "vae_kernel_avg_ms are the spans of kernels that share the CUs with each other; "
                                       "kernels_alone holds each kernel's own duration (a few steps on one stream)")
            if alone is not None:
                res["roofline"]["kernels_alone"] = alone
        tr = res["roofline"]["traffic"]
        if tr and lstm_avg_s > 0:
            # the other roofline, for the record: this path sits at a fraction of a percent of HBM
            res["roofline"]["hbm_achieved_GBps"] = tr / lstm_avg_s / 1e9
            res["roofline"]["hbm_frac_of_8TBps"] = tr / lstm_avg_s / 8e12
        if single is not None:
            res["single_gpu_same_workload"] = single
        if alt is not None:
            res["alt_arithmetic"] = alt
        res["library"] = library_info(sc)
        if world == 1 and default_workload and not args.no_others:
            # every other BASELINE config, and the host-pointer drop-in path, observed in this same run
            plan = {"c5": (5, 1, 4e6), "c3": (40, 5, 0), "c4": (8, 2, 0), "c1": (300, 20, 0), "c2": (20, 3, 0)}
            res["other_configs"] = [side_config(n, *plan[n][:2], local, plan[n][2])
                                    for n in ("c5", "c3", "c4", "c1", "c2") if n != wl["name"]]
            res["host_path"] = [host_path(n, c, local) for n, c in (("c2", 20), ("c4", 10))]
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    sc.close()


if __name__ == "__main__":
    main()
