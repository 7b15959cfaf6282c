"""estimator.predict latency against batch size, for the three kinds of Q-network (univariate H=128,
WaDi-shaped I=124/H=64, multivariate H=128): where the matrix-vector kernel hands over to the MFMA kernels."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from drlvae.scorer import WindowScorer
from test_gpu_parity import synth_lstm

sc = WindowScorer(0)
rng = np.random.default_rng(0)
for I, H in ((2, 128), (124, 64), (39, 128)):
    sc.set_lstm(synth_lstm(rng, I, H))
    for B in ([int(b) for b in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 5, 64, 256, 512, 768, 1024, 4096)):
        s = rng.random((B, 25, I), dtype=np.float32)
        for _ in range(5): sc.q_predict(s)
        t0 = time.perf_counter()
        for _ in range(50): sc.q_predict(s)
        print(f"I={I:3d} H={H:3d} batch {B:4d}: {(time.perf_counter() - t0) / 50 * 1e6:8.1f} us per call ({sc.lstm_variant})", flush=True)
