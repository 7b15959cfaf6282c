"""ORACLE (test infrastructure, not product code): the same CPU restatement as ref_path.py on
torch-CPU fp32 tensors (MKL / oneDNN threads), used ONLY as the second leg of ``bench.py``'s
``cpu_baseline`` (SURVEY.md section 8d names torch-CPU beside numpy) and checked against ref_path.py
by tests/test_oracle.py.  Same pinning status as ref_path.py; each function cites the reference lines
it follows.  Nothing under dynamic-reward-rl-vae_amd/ imports this module.
"""
from __future__ import annotations

import numpy as np
import torch


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def vae_err(X, vae):
    """Keras Dense chain, mean path, mean squared reconstruction error per row
    (reference myasp-A1.py:87-101, 140-141; myasp-wadi.py:101-117, 145-146)."""
    h = X
    ks, bs, acts = vae["kernels"], vae["biases"], vae["acts"]
    skip = vae["logvar_layer"]
    for l in range(len(ks)):
        if l == skip:
            continue
        h = h @ _t(ks[l]) + _t(bs[l])
        if acts[l] == 1:
            h = torch.relu(h)
        elif acts[l] == 2:
            h = torch.sigmoid(h)
    return torch.mean((h - X) ** 2, dim=1)


def lstm_q(S, lstm):
    """TF1 LSTMCell(forget_bias) under static_rnn + linear head (reference myasp-A1.py:176-179):
    z = [x_t, h].K + b, gates i|j|f|o, c = sig(f+fb)*c + sig(i)*tanh(j), h = sig(o)*tanh(c)."""
    K, b = _t(lstm["kernel"]), _t(lstm["bias"])
    H = K.shape[1] // 4
    B, W, I = S.shape
    fb = float(lstm.get("forget_bias", 1.0))
    h = torch.zeros(B, H)
    c = torch.zeros(B, H)
    for t in range(W):
        z = torch.cat([S[:, t, :], h], dim=1) @ K + b
        i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
        c = torch.sigmoid(f + fb) * c + torch.sigmoid(i) * torch.tanh(j)
        h = torch.sigmoid(o) * torch.tanh(c)
    return h @ _t(lstm["w_out"]) + _t(lstm["b_out"])


@torch.no_grad()
def score_series_univariate(v, W, vae, lstm, flag_mode="quirk"):
    """ref_path.score_series_univariate on torch-CPU (windows by unfold, flags as there)."""
    v = _t(v)
    K = v.shape[0] - W
    X = v[:-1].unfold(0, W, 1)[:K]                     # v[k .. k+W-1]   (reference myasp-A1.py:139)
    vals = v[1:].unfold(0, W, 1)[:K]                   # v[k+1 .. k+W]   (myasp-A1.py:120-133)
    S = torch.zeros(K, W, 2)
    S[:, :, 0] = vals
    if flag_mode == "quirk":
        kk = torch.arange(min(K, W))
        S[kk, W - 1 - kk, 1] = 1.0
    q = lstm_q(S, lstm)
    return {"err": vae_err(X, vae).numpy(), "q": q.numpy(), "action": (q[:, 1] > q[:, 0]).to(torch.int32).numpy()}


@torch.no_grad()
def score_series_multivariate(x, W, vae, lstm, center, scale, clip=10.0):
    """ref_path.score_series_multivariate on torch-CPU (reference myasp-wadi.py:132-149, 248-254, 326-328)."""
    x = _t(x)
    N, F = x.shape
    K = N - W
    win = x.unfold(0, W, 1).permute(0, 2, 1)           # (N-W+1, W, F): rows i..i+W-1
    Xs = win[1:K + 1].reshape(K, W * F).clone()
    Xs -= torch.from_numpy(np.asarray(center, np.float32))        # baseline arithmetic: float32 parameters
    Xs /= torch.from_numpy(np.asarray(scale, np.float32))
    Xs.clamp_(-clip, clip)
    err = vae_err(Xs, vae)
    raw = win[1:K + 1]
    S = torch.zeros(K, 2, W, F + 1)
    S[:, 0, :, :F] = raw
    S[:, 1, :, :F] = raw
    S[:, 1, :, F] = 1.0
    q = lstm_q(S.reshape(2 * K, W, F + 1), lstm).reshape(K, 2, -1)
    return {"err": err.numpy(), "q": q.numpy(), "action": (q[:, 1, 1] > q[:, 0, 0]).to(torch.int32).numpy()}
