#!/usr/bin/env python3
"""Diagnostic (needs a -DDRL_CLK_STAMPS build, DRLVAE_LIB=...): the shader clock the chip holds while an LSTM kernel
runs = s_memtime ticks / s_memrealtime ticks x 100 MHz between entry and exit of workgroup 0, after seconds of
back-to-back launches (MI355X_MICROARCH.md, DVFS give-back, item 6).  usage: kernel_clock.py <workload> [windows]
The stamps are those of the LAST stamped kernel of the call (C2/C5: lstm_h128i2; C3: lstm_rec2; C4: lstm_rec)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dynamic-reward-rl-vae_amd")]
import bench
from drlvae.scorer import WindowScorer
name = sys.argv[1]
wl = bench.workload(name, 0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
sc = WindowScorer(0)
sc.set_lstm(wl["lstm"])
kw = {k: v for k, v in wl["kw"].items() if not k.startswith("vae")}
t_end = time.time() + 2.5
n = 0
while time.time() < t_end:
    sc.score(wl["series"], wl["W"], wl["offsets"], want=("q",), **kw)
    n += 1
buf = (C.c_ulonglong * 4)()
sc._lib.drl_debug_clock.argtypes = [C.c_void_p]
assert sc._lib.drl_debug_clock(buf) == 0
t0, r0, t1, r1 = [int(x) for x in buf]
print(f"{name}: {sc.lstm_kernels}  workgroup 0: {(r1 - r0) / 100:.1f} us, {t1 - t0} shader cycles -> {(t1 - t0) / (r1 - r0) * 100:.0f} MHz  ({n} calls)")
