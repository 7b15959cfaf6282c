import sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "./dynamic-reward-rl-vae_amd")
import bench
from drlvae.scorer import WindowScorer
for name in ("c3", "c4"):
    wl = bench.workload(name, 0, 0)
    rng = np.random.default_rng(1)
    lstm = bench.make_lstm(rng, wl["F"] + 1, 128)
    sc = WindowScorer(0); sc.set_vae(wl["vae"]); sc.set_lstm(lstm)
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(wl["series"]).to(dev)
    K = wl["K"]
    out = {"q": torch.empty((K, 2, 2), device=dev), "action": torch.empty(K, dtype=torch.int32, device=dev)}
    torch.cuda.synchronize()
    kw = dict(wl["kw"])
    for _ in range(2): sc.score_dev(x, wl["W"], wl["offsets"], out, **kw)
    sc.synchronize(); sc.profile(True); sc.profile_reset()
    for _ in range(3): sc.score_dev(x, wl["W"], wl["offsets"], out, **kw)
    sc.synchronize(); ms, n = sc.profile_read(0)
    I = wl["F"] + 1
    fl = 2 * (2 * 25 * (I + 128) * 512 + 2 * 128 * 2) * K
    print(name, "H=128", sc.lstm_variant, f"{ms/n:.3f} ms/launch", f"{fl/(ms/n*1e-3)/1e12:.1f} TFLOP/s algorithmic", f"{K/(ms/n*1e-3)/1e6:.2f} Mwin/s (LSTM only)")
    sc.close()
