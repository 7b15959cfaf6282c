"""Fits T(tile) = a + b*W for the univariate H=128 LSTM kernel: one series of 256*R*64 windows
(every CU gets exactly R tiles), several window lengths.  usage: python tools/fit_tile_cost.py [R]"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dynamic-reward-rl-vae_amd"))
from drlvae.scorer import WindowScorer

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
H, I = 128, 2
lim = np.sqrt(6.0 / (I + H + 4 * H))
lstm = {"kernel": rng.uniform(-lim, lim, (I + H, 4 * H)).astype(np.float32), "bias": np.zeros(4 * H, np.float32),
        "w_out": rng.normal(0, 1, (H, 2)).astype(np.float32), "b_out": np.zeros(2, np.float32)}
sc = WindowScorer(0)
sc.set_lstm(lstm)
dev = torch.device("cuda", 0)
K = 256 * R * 64
res = []
for W in (13, 25, 50, 100, 150):
    x = torch.rand(K + W, device=dev)
    q = torch.empty((K, 1, 2), device=dev)
    torch.cuda.synchronize()
    out = {"q": q}
    for _ in range(2):
        sc.score_dev(x, W, np.array([0, K + W]), out)
    sc.synchronize(); sc.profile(True); sc.profile_reset()
    for _ in range(5):
        sc.score_dev(x, W, np.array([0, K + W]), out)
    sc.synchronize()
    ms, n = sc.profile_read(0); sc.profile(False)
    t = ms / n * 1e3 / R
    res.append((W, t))
    print(f"W={W:4d}  launch {ms/n:8.3f} ms  per tile {t:8.1f} us  per step {t/W:6.2f} us", flush=True)
A = np.array([[1, w] for w, _ in res]); y = np.array([t for _, t in res])
(a, b), *_ = np.linalg.lstsq(A, y, rcond=None)
print(f"fit: per tile = {a:.1f} us + {b:.2f} us * W   (MFMA floor per step at 2.39 GHz: 13.9 us)")
