#!/usr/bin/env python3
# How fast does an 85 MB host buffer reach the device?  pageable hipMemcpy, pinned hipMemcpy, and a threaded copy into
# pinned staging buffers followed by the DMA (what drl_score_batch could do for pageable callers).
import time, numpy as np, torch, concurrent.futures as cf
n = 172801 * 123
x = np.random.default_rng(0).random(n, dtype=np.float32)
dev = torch.device("cuda", 0)
d = torch.empty(n, dtype=torch.float32, device=dev)
def t(f, reps=8):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.time(); f(); torch.cuda.synchronize(); ts.append(time.time() - t0)
    return np.median(ts)
tx = torch.from_numpy(x)
dt = t(lambda: d.copy_(tx)); print(f"pageable -> device: {dt*1e3:.2f} ms  {x.nbytes/dt/1e9:.1f} GB/s")
px = torch.empty(n, dtype=torch.float32).pin_memory(); px.copy_(tx)
dt = t(lambda: d.copy_(px, non_blocking=True)); print(f"pinned   -> device: {dt*1e3:.2f} ms  {x.nbytes/dt/1e9:.1f} GB/s")
pn = px.numpy()
for nt in (1, 2, 4, 8, 16):
    pool = cf.ThreadPoolExecutor(nt)
    cuts = np.linspace(0, n, nt + 1).astype(np.int64)
    def cp():
        list(pool.map(lambda i: np.copyto(pn[cuts[i]:cuts[i+1]], x[cuts[i]:cuts[i+1]]), range(nt)))
    dt = t(cp); print(f"host memcpy pageable -> pinned, {nt:2d} threads: {dt*1e3:.2f} ms  {x.nbytes/dt/1e9:.1f} GB/s")
