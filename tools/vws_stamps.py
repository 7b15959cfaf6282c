#!/usr/bin/env python3
"""Diagnostic (needs a -DVWS_STAMPS build, DRLVAE_LIB=...): s_memtime stamps of vae_wide16s_kernel's layer-0 slabs for
wave 0 of workgroup 0: transform + LDS reads issued + first-half MFMAs | reads, stores, requests + second half | barrier."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")]
import bench
from drlvae.scorer import WindowScorer
wl = bench.workload(sys.argv[2] if len(sys.argv) > 2 else "c4", 0, int(sys.argv[1]) if len(sys.argv) > 1 else 32768)
sc = WindowScorer(0)
sc.set_vae(wl["vae"])
sc.set_knob("vae_s", 1)
kw = {k: v for k, v in wl["kw"].items() if k.startswith("vae")}
import time
if os.environ.get("VWS_EPS"):                               # the sampled-z path: the logvar head in the middle layers
    Kw = int(sum(max(0, b - a - wl["W"]) for a, b in zip(wl["offsets"][:-1], wl["offsets"][1:])))
    kw["eps"] = np.random.default_rng(0).normal(size=(Kw, 10)).astype(np.float32)
t_end = time.time() + float(os.environ.get("VWS_WARM_S", "2.5"))      # the clock the chip holds shows after seconds of load
while time.time() < t_end:
    sc.score(wl["series"], wl["W"], wl["offsets"], want=("err",), **kw)
assert sc.vae_variant == "wide16s", sc.vae_variant
buf = (C.c_ulonglong * 256)()
sc._lib.drl_debug_stamps_s.argtypes = [C.c_void_p, C.c_int]
assert sc._lib.drl_debug_stamps_s(buf, 256) == 0
t = np.array(buf, dtype=np.uint64).reshape(64, 4).astype(np.int64)
for it in range(2, 8):
    r = t[it]
    print(f"slab {it:2d}: first half {r[1]-r[0]:5d}  second half {r[2]-r[1]:5d}  barrier {r[3]-r[2]:5d}  | slab total {t[it+1][0]-r[0]:5d}")
print("mean cycles per slab", float(np.diff(t[2:59, 0]).mean()))
m = np.array(buf, dtype=np.uint64).astype(np.int64)[240:247]
names = ["region staged", "layer-0 prologue", "layer-0 loop", "act + middle layers", "last-layer prologue", "last-layer loop"]
for i, nm in enumerate(names):
    print(f"{nm:22s} {m[i+1]-m[i]:8d} cycles")
print("total", m[6] - m[0])
rt = np.array(buf, dtype=np.uint64).astype(np.int64)[248:250]
print(f"in-kernel clock of workgroup 0: {(m[6] - m[0]) / (rt[1] - rt[0]) * 100:.0f} MHz  (s_memtime ticks / s_memrealtime ticks x 100 MHz; {(rt[1]-rt[0]) / 100:.1f} us)")
