"""Latency of the reference's per-call surface on this library: estimator.predict(state) at batch 1,
2 (the two candidate next-states of env.step) and 512 (2 x 256 replay next-states of a DQN update),
vae.predict at batch 1, host numpy buffers in and out (synchronous calls, as sess.run is)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd"))
import numpy as np
import bench
from drlvae.scorer import WindowScorer

wl = bench.workload("c1", 0, 0)
sc = WindowScorer(0)
sc.set_vae(wl["vae"]); sc.set_lstm(wl["lstm"])
rng = np.random.default_rng(0)
for B in (1, 2, 512):
    s = rng.random((B, 25, 2), dtype=np.float32)
    for _ in range(20): sc.q_predict(s)
    t0 = time.perf_counter()
    for _ in range(200): sc.q_predict(s)
    print(f"estimator.predict batch {B:4d}: {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us per call ({sc.lstm_variant})")
x = rng.random((1, 25), dtype=np.float32)
for _ in range(20): sc.vae_predict(x)
t0 = time.perf_counter()
for _ in range(200): sc.vae_predict(x)
print(f"vae.predict batch 1: {(time.perf_counter() - t0) / 200 * 1e6:8.1f} us per call")
# where the time of a batch-1 call goes: the kernel itself (HIP events around the launch)
s = rng.random((1, 25, 2), dtype=np.float32)
sc.profile(True); sc.profile_reset()
for _ in range(200): sc.q_predict(s)
ms, n = sc.profile_read(0)
print(f"  of which the matrix-vector kernel: {ms / max(n, 1) * 1e3:.1f} us (HIP events)")
sc.profile_reset()
for _ in range(200): sc.vae_predict(x)
ms, n = sc.profile_read(1)
print(f"  vae kernel: {ms / max(n, 1) * 1e3:.1f} us")
sc.profile(False)
