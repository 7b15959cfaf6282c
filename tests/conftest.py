import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-reward-rl-vae_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_model_npz(name):
    """(lstm dict, vae dict) from a tests/golden/*.npz written by make_golden.py."""
    z = np.load(os.path.join(GOLDEN, name))
    lstm = {k[5:]: z[k] for k in z.files if k.startswith("lstm_")}
    vae = None
    if "vae_acts" in z.files:
        n = len(z["vae_acts"])
        vae = {"kernels": [z[f"vae_k{i}"] for i in range(n)],
               "biases": [z[f"vae_b{i}"] for i in range(n)],
               "acts": [int(a) for a in z["vae_acts"]],
               "mu_layer": int(z["vae_mu_layer"]),
               "logvar_layer": int(z["vae_logvar_layer"]),
               "logvar_clip": float(z["vae_logvar_clip"])}
    return lstm, vae


def split_series(z):
    off = z["offsets"]
    return [z["values"][off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.fixture(scope="session")
def a1():
    z = np.load(os.path.join(GOLDEN, "a1_series.npz"))
    return {"series": split_series(z), "names": [str(n) for n in z["names"]],
            "anomaly": [z["anomaly"][z["offsets"][i]:z["offsets"][i + 1]] for i in range(len(z["offsets"]) - 1)],
            "raw3": [z["raw3"][z["raw3_offsets"][i]:z["raw3_offsets"][i + 1]] for i in range(3)]}


@pytest.fixture(scope="session")
def ckpt_a1():
    return load_model_npz("ckpt_a1.npz")


@pytest.fixture(scope="session")
def ckpt_a2():
    return load_model_npz("ckpt_a2.npz")
