"""Numerics check (CPU, numpy): LSTM-DQN with the recurrent product emulated as a 3-term f16 split
(U_hi.h_hi + U_hi.h_lo + U_lo.h_hi, fp32 accumulate) against the fp64 oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from conftest import load_model_npz
from oracle import ref_path as orc

def split16(x):
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)

def lstm_split(states, lstm, terms=3):
    s = np.asarray(states, np.float32)
    B, W, I = s.shape
    K = lstm["kernel"].astype(np.float64); H = K.shape[1] // 4
    L2E = 1.4426950408889634
    sc = np.concatenate([np.full(H, -L2E), np.full(H, -2 * L2E), np.full(H, -L2E), np.full(H, -L2E)])
    Kp = (K * sc).astype(np.float32)
    bp = ((lstm["bias"].astype(np.float64) + np.concatenate([np.zeros(2 * H), np.ones(H), np.zeros(H)])) * sc).astype(np.float32)
    Wx, U = Kp[:I], Kp[I:]
    Uhi, Ulo = split16(U)
    h = np.zeros((B, H), np.float32); c = np.zeros((B, H), np.float32)
    for t in range(W):
        z = s[:, t, :] @ Wx + bp
        hhi, hlo = split16(h)
        z = z + hhi @ Uhi
        if terms >= 2: z = z + hlo @ Uhi
        if terms >= 3: z = z + hhi @ Ulo
        z = z.astype(np.float32)
        ei, ej, ef, eo = [np.exp2(z[:, k * H:(k + 1) * H]) for k in range(4)]
        si, tj, sf, so = 1 / (1 + ei), 2 / (1 + ej) - 1, 1 / (1 + ef), 1 / (1 + eo)
        c = (sf * c + si * tj).astype(np.float32)
        h = (so * (2 / (1 + np.exp2(np.float32(-2 * L2E) * c)) - 1)).astype(np.float32)
    return h @ lstm["w_out"] + lstm["b_out"]

def rel(got, want):
    return float(np.max(np.abs(got - want) / np.maximum(np.abs(want), np.abs(want).max(axis=-1, keepdims=True))))

lstm, _ = load_model_npz("ckpt_a1.npz")
z = np.load(os.path.join(ROOT, "tests/golden/a1_series.npz"))
v = z["values"][:1420]
for W in (25, 50):
    S = orc.states_list_univariate(v, W)
    want = orc.lstm_q(S, lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], dtype=np.float64)
    f32 = orc.lstm_q(S, lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], dtype=np.float32)
    print(f"A1 ckpt W={W}: fp32 oracle rel {rel(f32, want):.2e}", *[f"| {t}-term split rel {rel(lstm_split(S, lstm, t), want):.2e}" for t in (1, 2, 3)])
rng = np.random.default_rng(0)
for trial in range(3):
    lim = np.sqrt(6.0 / (130 + 512)) * (1 + 2 * trial)
    l2 = {"kernel": rng.uniform(-lim, lim, (130, 512)).astype(np.float32), "bias": rng.uniform(-.1, .1, 512).astype(np.float32),
          "w_out": rng.normal(0, 1, (128, 2)).astype(np.float32), "b_out": rng.normal(0, 1, 2).astype(np.float32)}
    S = orc.states_list_univariate(rng.random(600).astype(np.float32), 50)
    want = orc.lstm_q(S, l2["kernel"], l2["bias"], l2["w_out"], l2["b_out"], dtype=np.float64)
    f32 = orc.lstm_q(S, l2["kernel"], l2["bias"], l2["w_out"], l2["b_out"], dtype=np.float32)
    print(f"synthetic x{1+2*trial} W=50: fp32 oracle rel {rel(f32, want):.2e} | 3-term split rel {rel(lstm_split(S, l2, 3), want):.2e} | 2-term {rel(lstm_split(S, l2, 2), want):.2e}")
