"""vae.predict latency for wide windows (SMD D=950, WaDi D=3075), explicit batches as the reference
passes them (myasp-wadi.py:143-146 uses chunks of 64 windows)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drlvae.scorer import WindowScorer
from test_gpu_parity import synth_vae

sc = WindowScorer(0)
rng = np.random.default_rng(0)
for D in (950, 3075):
    sc.set_vae(synth_vae(rng, [D, 128, 10, 128, D], 1))
    for B in (1, 64, 1024):
        x = rng.random((B, D), dtype=np.float32)
        for _ in range(3): sc.vae_predict(x)
        t0 = time.perf_counter()
        for _ in range(20): sc.vae_predict(x)
        print(f"D={D:4d} batch {B:4d}: {(time.perf_counter() - t0) / 20 * 1e6:9.1f} us per call", flush=True)
