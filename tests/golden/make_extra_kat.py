#!/usr/bin/env python3
"""A third soft known-answer fixture (the checkpoint sweep): the LSTM weights of the
reference's A1 LP200/AL10000 run (its committed performance file is reproduced 5 of 6 episodes
exactly) -> ckpt_a1_al10000.npz + soft_kat_extra.json.  Reads reference DATA only (the TF1 bundle
with this repo's own reader, the performance text file); run in the build container."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "dynamic-reward-rl-vae_amd"))
sys.path.insert(0, HERE)
from drlvae.formats.tf_bundle import TFBundle, load_q_estimator  # noqa: E402
from make_golden import parse_perf, save_model_npz  # noqa: E402  (imports the reference's env.py as make_golden does)

REF = "/root/reference"
RUN = "A1_LP_1500init_warmup_h128_b256_300ep_num_LP200_num_AL10000_d0.96"

prefix = os.path.join(REF, "exp", RUN, "checkpoints", "model")
lstm = load_q_estimator(prefix, "qlearn")
b = TFBundle(prefix)
for nm in ("qlearn/rnn/lstm_cell/kernel", "qlearn/rnn/lstm_cell/bias"):
    b.get(nm, verify_crc=True)
save_model_npz(os.path.join(HERE, "ckpt_a1_al10000.npz"), lstm, None)
names = [str(n) for n in np.load(os.path.join(HERE, "a1_series.npz"))["names"]]
first = int(np.ceil(len(names) * 0.9))
json.dump({"a1_al10000": {"run": RUN, "global_step": int(b.get("global_step")), "first_index": first,
                          "files": names[first:first + 6],
                          "perf": parse_perf(os.path.join(REF, "exp", RUN + "performance.txt"))}},
          open(os.path.join(HERE, "soft_kat_extra.json"), "w"), indent=1)
print("written", RUN, "global_step", int(b.get("global_step")))
