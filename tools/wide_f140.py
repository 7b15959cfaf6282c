#!/usr/bin/env python3
# Times the wide-window VAE kernels on a shape whose region leaves room for ONE workgroup per CU (F = 140, W = 25):
# vae_s = 1 (vae_wide16s_kernel, alone on its CU) against 0 (vae_wide16_kernel).
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd"), os.path.join(ROOT, "tests")]
from drlvae.scorer import WindowScorer
rng = np.random.default_rng(1)
W, F, N = 25, 140, 131072 + 25
D = W * F
lim = lambda a, b: np.sqrt(6.0 / (a + b))
dims = [(D, 128), (128, 10), (128, 10), (10, 128), (128, D)]
vae = {"kernels": [rng.uniform(-lim(a, b), lim(a, b), (a, b)).astype(np.float32) for a, b in dims],
       "biases": [rng.uniform(-0.05, 0.05, b).astype(np.float32) for a, b in dims], "acts": [1, 0, 0, 1, 2],
       "mu_layer": 1, "logvar_layer": 2, "logvar_clip": 10.0}
x = rng.random((N, F), dtype=np.float32)
sc = WindowScorer(0); sc.set_vae(vae)
dev = torch.device("cuda", 0)
series = torch.from_numpy(x).to(dev)
K = N - W
out = {"err": torch.empty(K, dtype=torch.float32, device=dev)}
kw = dict(vae_lag=0, vae_center=rng.uniform(0.3, 0.7, D), vae_scale=rng.uniform(0.05, 0.6, D), vae_clip=10.0)
torch.cuda.synchronize()
for knob in (1, 0):
    sc.set_knob("vae_s", knob)
    for _ in range(10): sc.score_dev(series, W, [0, N], out, **kw)
    sc.synchronize(); ts = []
    for _ in range(10):
        t = time.time(); sc.score_dev(series, W, [0, N], out, **kw); sc.synchronize(); ts.append((time.time() - t) * 1e3)
    print(f"vae_s={knob} {sc.vae_variant:10s} {K} windows of {D}: median {np.median(ts):.3f} ms")
