#!/usr/bin/env python3
# Times the sliding-window VAE error with and without eps (the reference's sampled z) on C3 and C4, device-resident.
import os, sys, time, numpy as np, torch
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")]
import bench
from drlvae.scorer import WindowScorer
for name in ("c3", "c4"):
    wl = bench.workload(name, 0, 0)
    sc = WindowScorer(0); sc.set_vae(wl["vae"])
    kw = {k: v for k, v in wl["kw"].items() if k.startswith("vae")}
    K = int(sum(max(0, b - a - wl["W"]) for a, b in zip(wl["offsets"][:-1], wl["offsets"][1:])))
    latent = 10
    dev = torch.device("cuda", 0)
    series = torch.from_numpy(wl["series"]).to(dev)
    out = {"err": torch.empty(K, dtype=torch.float32, device=dev)}
    eps = torch.randn((K, latent), device=dev)
    for e in (None, eps):
        for _ in range(30): sc.score_dev(series, wl["W"], wl["offsets"], out, eps=e, **kw)
        torch.cuda.synchronize(); sc.synchronize(); t = time.time()
        ts = []
        for _ in range(10):
            t1 = time.time(); sc.score_dev(series, wl["W"], wl["offsets"], out, eps=e, **kw); sc.synchronize()
            ts.append((time.time() - t1) * 1e3)
        print(name, K, "eps" if e is not None else "mean", sc.vae_variant, f"median {np.median(ts):.3f} ms per synchronous call (min {min(ts):.3f}, max {max(ts):.3f})")
