#!/usr/bin/env python3
# Times the sliding-window VAE error on a WaDi-shaped stream (F = 123, W = 25, 172,776 windows) for several hidden widths:
# 97..128 take the wide kernels, narrower ones the chain kernel.
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")]
from drlvae.scorer import WindowScorer
rng = np.random.default_rng(1)
W, F, N = 25, 123, 172801
D = W * F
lim = lambda a, b: np.sqrt(6.0 / (a + b))
x = rng.random((N, F), dtype=np.float32)
dev = torch.device("cuda", 0)
series = torch.from_numpy(x).to(dev)
K = N - W
out = {"err": torch.empty(K, dtype=torch.float32, device=dev)}
kw = dict(vae_lag=0, vae_center=rng.uniform(0.3, 0.7, D), vae_scale=rng.uniform(0.05, 0.6, D), vae_clip=10.0)
sc = WindowScorer(0)
torch.cuda.synchronize()
for hid in (32, 64, 96, 112, 128):
    dims = [(D, hid), (hid, 10), (hid, 10), (10, hid), (hid, D)]
    vae = {"kernels": [rng.uniform(-lim(a, b), lim(a, b), (a, b)).astype(np.float32) for a, b in dims],
           "biases": [rng.uniform(-0.05, 0.05, b).astype(np.float32) for a, b in dims], "acts": [1, 0, 0, 1, 2],
           "mu_layer": 1, "logvar_layer": 2, "logvar_clip": 10.0}
    sc.set_vae(vae)
    for _ in range(5): sc.score_dev(series, W, [0, N], out, **kw)
    sc.synchronize(); ts = []
    for _ in range(8):
        t = time.time(); sc.score_dev(series, W, [0, N], out, **kw); sc.synchronize(); ts.append((time.time() - t) * 1e3)
    fl = 2.0 * (2 * D * hid + 2 * hid * 10) * K
    print(f"hidden {hid:4d}: {sc.vae_variant:8s} median {np.median(ts):.3f} ms  {fl / np.median(ts) / 1e9:.1f} TFLOP/s")
