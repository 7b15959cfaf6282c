"""PCIe-inclusive rate of the host-pointer entry point (drl_score_batch: pageable numpy buffers in,
numpy buffers out) on a bench workload, next to the HBM-resident rate bench.py reports as `value`.
usage: python tools/host_path_rate.py [c2|c3|c4]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd"))
import bench
from drlvae.scorer import WindowScorer

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
wl = bench.workload(name, 0, 0)
sc = WindowScorer(0)
sc.set_vae(wl["vae"]); sc.set_lstm(wl["lstm"])
kw = dict(wl["kw"])
if os.environ.get("HOST_PATH_PINNED"):                      # what the path would do with a pinned caller buffer
    import torch
    pinned = torch.from_numpy(wl["series"]).pin_memory()
    wl["series"] = pinned.numpy()
for _ in range(3):
    sc.score(wl["series"], wl["W"], wl["offsets"], **kw)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    out = sc.score(wl["series"], wl["W"], wl["offsets"], **kw)
dt = (time.perf_counter() - t0) / n
print(f"{name}: host-pointer path {wl['K'] / dt / 1e6:.2f} M windows/s ({dt * 1e3:.3f} ms per call, "
      f"{wl['series'].nbytes / 1e6:.2f} MB in, {sum(v.nbytes for v in out.values()) / 1e6:.2f} MB out)")
