#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the sources libdrlvae.so is built from, in a fixed order.  csrc/Makefile
compiles it into the library (drl_version); bench.py and tests/test_abi.py compare the two, so a prebuilt
library that no longer matches the sources beside it is noticed."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["include/drlvae.h"] + ["dynamic-reward-rl-vae_amd/csrc/" + f for f in
                                ("drlvae.hip", "kernels.hip.h", "lstm_f16x3.hip.h", "lstm_w16.hip.h", "vae_wide16.hip.h", "vae_wide16pp.hip.h", "vae_wide16s.hip.h")]


def src_hash(root=ROOT):
    h = hashlib.sha256()
    for f in FILES:
        h.update(f.encode() + b"\0")
        with open(os.path.join(root, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(src_hash())
