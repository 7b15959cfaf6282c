"""Time of the sequential validator rollout (drl_rollout_batch), both exact designs, on the committed
validation series: the fixed point over full-parallel scoring passes and the persistent per-series kernel.
Run on the GPU box; the output of this script is kept in profiles/rNN/rollout_time.txt."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_model_npz, GOLDEN
from drlvae.scorer import WindowScorer

W = 25
sc = WindowScorer(0)


def timed(values, offsets, label):
    res = {}
    for mode in ("fixed_point_full_passes", "fixed_point", "sequential", "auto"):
        sc.set_rollout_mode("fixed_point" if mode.startswith("fixed") else mode)
        sc.set_knob("rollout_dense", 1 if mode == "fixed_point_full_passes" else 0)
        for _ in range(2):
            sc.rollout(values, W, offsets)
        t0 = time.perf_counter()
        for _ in range(5):
            a = sc.rollout(values, W, offsets)
        res[mode] = ((time.perf_counter() - t0) / 5, a, sc.last_rollout_passes)
        sc.set_knob("rollout_dense", 0)
    (tf, af, pf), (ts, as_, _), (ta, aa, pa) = res["fixed_point"], res["sequential"], res["auto"]
    td, ad, pd_ = res["fixed_point_full_passes"]
    assert np.array_equal(af, as_) and np.array_equal(af, aa) and np.array_equal(af, ad) and pd_ == pf
    n = 1 if offsets is None else len(offsets) - 1
    print(f"{label}: {n} series, {len(af)} cursors, {int(af.sum())} alarms | fixed point {pf} passes {tf * 1e3:.2f} ms "
          f"({tf / pf * 1e6:.0f} us per pass; with full passes only {td * 1e3:.2f} ms) | sequential kernel {ts * 1e3:.2f} ms ({ts / max(1, len(af) if n == 1 else max(np.diff(offsets)) - W) * 1e6:.1f} us per cursor of the longest series)"
          f" | faster: {'sequential' if ts < tf else 'fixed point'} x{max(tf, ts) / min(tf, ts):.1f}"
          f" | auto {ta * 1e3:.2f} ms ({pa} passes{', then the sequential kernel' if sc.lstm_variant == 'rollout_seq' else ''})")


lstm2, _ = load_model_npz("ckpt_a2.npz")
z2 = np.load(os.path.join(GOLDEN, "a2_val_series.npz"))
sc.set_lstm(lstm2)
timed(z2["values"], z2["offsets"], "A2 validation set (ckpt A2 LP200/AL100)")
timed(z2["values"][z2["offsets"][0]:z2["offsets"][1]], None, "A2 first validation series alone")

lstm1, _ = load_model_npz("ckpt_a1.npz")
z1 = np.load(os.path.join(GOLDEN, "a1_series.npz"))
sc.set_lstm(lstm1)
off = z1["offsets"]
names = [str(n) for n in z1["names"]]
i7 = names.index("real_7.csv")
timed(z1["values"][off[i7]:off[i7 + 1]], None, "A1 real_7 alone (anomaly-dense: the reference's run predicts 44 positives)")
val = list(range(61, 67))
vals = np.concatenate([z1["values"][off[i]:off[i + 1]] for i in val])
voff = np.concatenate([[0], np.cumsum([off[i + 1] - off[i] for i in val])])
timed(vals, voff, "A1 six validation series (#61-66)")
timed(z1["values"], off, "all 67 A1 series")
sc.close()
