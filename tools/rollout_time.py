"""Time of the sequential validator rollout (drl_rollout_batch) on the committed validation series."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_model_npz, GOLDEN
from drlvae.scorer import WindowScorer
lstm, _ = load_model_npz("ckpt_a2.npz")
z = np.load(os.path.join(GOLDEN, "a2_val_series.npz"))
sc = WindowScorer(0); sc.set_lstm(lstm)
for _ in range(2): sc.rollout(z["values"], 25, z["offsets"])
t0 = time.perf_counter()
for _ in range(5): a = sc.rollout(z["values"], 25, z["offsets"])
dt = (time.perf_counter() - t0) / 5
K = len(a)
print(f"A2 validation set: {len(z['offsets'])-1} series, {K} cursors, {sc.last_rollout_passes} passes, {dt*1e3:.2f} ms per rollout "
      f"({dt/sc.last_rollout_passes*1e6:.0f} us per pass)")
one = z["values"][z["offsets"][0]:z["offsets"][1]]
for _ in range(2): sc.rollout(one, 25)
t0 = time.perf_counter()
for _ in range(5): sc.rollout(one, 25)
dt = (time.perf_counter() - t0) / 5
print(f"one series ({len(one)-25} cursors): {sc.last_rollout_passes} passes, {dt*1e3:.2f} ms")
