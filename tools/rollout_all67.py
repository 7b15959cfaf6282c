"""drl_rollout_batch on all 67 A1 series, fixed point, a few times: the command behind the rollout kernel trace."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_model_npz, GOLDEN
from drlvae.scorer import WindowScorer
sc = WindowScorer(0)
lstm1, _ = load_model_npz("ckpt_a1.npz")
z1 = np.load(os.path.join(GOLDEN, "a1_series.npz"))
sc.set_lstm(lstm1)
sc.set_rollout_mode(sys.argv[1] if len(sys.argv) > 1 else "fixed_point")
for _ in range(2):
    sc.rollout(z1["values"], 25, z1["offsets"])
t0 = time.perf_counter()
for _ in range(5):
    a = sc.rollout(z1["values"], 25, z1["offsets"])
print(f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per rollout, {sc.last_rollout_passes} passes, {int(a.sum())} alarms")
sc.close()
