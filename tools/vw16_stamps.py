#!/usr/bin/env python3
"""Diagnostic (needs a -DVW16_STAMPS build, DRLVAE_LIB=...): s_memtime stamps of vae_wide16_kernel's layer-0 slabs for
wave 0 of workgroup 0: requests+LDS reads issued | MFMAs (+transform) | staging stores | barrier."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")]
import bench
from drlvae.scorer import WindowScorer
wl = bench.workload("c4", 0, int(sys.argv[1]) if len(sys.argv) > 1 else 32768)
sc = WindowScorer(0)
sc.set_vae(wl["vae"])
kw = {k: v for k, v in wl["kw"].items() if k.startswith("vae")}
for _ in range(3):
    sc.score(wl["series"], wl["W"], wl["offsets"], want=("err",), **kw)
buf = (C.c_ulonglong * 320)()
sc._lib.drl_debug_stamps16.argtypes = [C.c_void_p, C.c_int]
assert sc._lib.drl_debug_stamps16(buf, 320) == 0
t = np.array(buf, dtype=np.uint64).reshape(64, 5).astype(np.int64)
for it in range(2, 20):
    r = t[it]
    print(f"slab {it:2d}: loads+reads issued {r[1]-r[0]:5d}  mfma {r[2]-r[1]:5d}  stores {r[3]-r[2]:5d}  barrier {r[4]-r[3]:5d}  | slab total {t[it+1][0]-r[0]:5d}")
d = t[2:60, 0]
print("mean cycles per slab", float(np.diff(d).mean()))
