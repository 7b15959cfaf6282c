#!/usr/bin/env python3
"""Diagnostic (needs a -DVWP_STAMPS build, DRLVAE_LIB=...): prints the s_memtime stamps of the first half-steps of
vae_wide16pp_kernel's layer 0 for one wave of each group: start, end of work, after the barrier."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")]
import bench
from drlvae.scorer import WindowScorer
wl = bench.workload("c4", 0, 32768)
sc = WindowScorer(0)
sc.set_knob("vae_pp", 2)
sc.set_vae(wl["vae"])
kw = {k: v for k, v in wl["kw"].items() if k.startswith("vae")}
for _ in range(3):
    sc.score(wl["series"], wl["W"], wl["offsets"], want=("err",), **kw)
buf = (C.c_ulonglong * 384)()
sc._lib.drl_debug_stamps.argtypes = [C.c_void_p, C.c_int]
assert sc._lib.drl_debug_stamps(buf, 384) == 0
t = np.array(buf, dtype=np.uint64).reshape(2, 64, 3).astype(np.int64)
t0 = t[0, 0, 0]
for h in range(1, 24):
    a, b = t[0, h] - t0, t[1, h] - t0
    print(f"h={h:2d}  A: start {a[0]:7d} work {a[1]-a[0]:6d} barrier-wait {a[2]-a[1]:6d} | B: start {b[0]:7d} work {b[1]-b[0]:6d} barrier-wait {b[2]-b[1]:6d}")
m = np.array(buf, dtype=np.uint64).astype(np.int64)
for g in (0, 1):
    base = (g * 64 + 60) * 3 - 24
    mk = m[base:base + 5]
    print(f"group {'AB'[g]}: prologue (regions, slab 0) {mk[1]-mk[0]}  layer 0 {mk[2]-mk[1]}  middle {mk[3]-mk[2]}  last layer {mk[4]-mk[3]}  total {mk[4]-mk[0]} cycles")
