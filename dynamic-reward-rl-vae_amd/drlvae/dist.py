"""Multi-GPU form of the scoring path: one process per GPU, weights replicated, windows sharded.

Windows are independent units, so there is no data-path exchange; the only collective is the
all-gather of the per-window scores (16-24 B per window; RCCL over xGMI with backend "nccl", gloo
on CPU for the tests).  The reference is single-process; this module has no counterpart there.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n units: rank r gets [r*n//world, (r+1)*n//world)."""
    return (rank * n) // world, ((rank + 1) * n) // world


def shard_stream(n_samples: int, W: int, rank: int, world: int, vae_lag: int = 1):
    """Shard of one long series.  Window k (cursor k+W) needs samples k+1-vae_lag .. k+W, so the
    shard that owns windows [k_lo, k_hi) reads samples [k_lo, k_hi+W) — a (W-1)+lag sample halo
    shared with the next rank.  Scoring that slice as a series of its own yields exactly windows
    k_lo..k_hi-1, in order, EXCEPT for position-dependent flag modes (the get_states_list quirk
    marks windows k < W of the *series*), which is why callers pass ``k_lo`` on.

    Returns (sample_lo, sample_hi, k_lo, k_hi)."""
    K = max(0, n_samples - W)
    k_lo, k_hi = shard_range(K, rank, world)
    if k_hi <= k_lo:
        return 0, 0, k_lo, k_lo
    return k_lo, k_hi + W, k_lo, k_hi


def deal_series(lengths: Sequence[int], W: int, world: int) -> List[List[int]]:
    """Ragged batches: whole series are dealt to ranks, longest first onto the lightest rank
    (windows never cross a series, so a series is the natural unit).  Returns per rank the list
    of series indices in ascending order."""
    k = [max(0, int(n) - W) for n in lengths]
    load = [0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(k)), key=lambda i: (-k[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        out[r].append(i)
        load[r] += k[i]
    return [sorted(s) for s in out]


def allgather_varlen(local, counts: Sequence[int], group=None):
    """All-gather of per-rank tensors whose first dimension differs (counts[r] rows on rank r).
    Pads to the largest shard so that a single fixed-size all-gather moves everything."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    m = max(counts) if len(counts) else 0
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    buf = torch.empty((world, m) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf.view(world * m, *local.shape[1:]) if m else buf, pad, group=group)
    return [buf[r, : counts[r]] for r in range(world)]


def score_stream_sharded(score_fn: Callable[..., Dict], series: np.ndarray, W: int, *, vae_lag: int = 1,
                         flag_mode: str = "zeros", group=None, device=None, **kw) -> Dict:
    """Scores one long series across all ranks of ``group`` and returns, on every rank, the full
    per-window result (identical to a single-process call).

    ``score_fn(series_slice, W, vae_lag=..., flag_mode=..., **kw) -> {"err","q","action","margin"}``
    is ``WindowScorer.score`` on a GPU rank.
    """
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    x = np.asarray(series, dtype=np.float32)
    n = x.shape[0]
    if flag_mode == "quirk" and world > 1:
        # the quirk flags depend on the window's position in the whole series: express them as
        # explicit per-sample flags, which shard like the samples do
        flags = np.zeros(n, np.float32)
        if n > W:
            flags[W] = 1.0
        kw = dict(kw, flags=flags)
        flag_mode = "explicit"
    counts = []
    for r in range(world):
        _, _, lo, hi = shard_stream(n, W, r, world, vae_lag)
        counts.append(hi - lo)
    s_lo, s_hi, k_lo, k_hi = shard_stream(n, W, rank, world, vae_lag)
    local_kw = dict(kw)
    if "flags" in local_kw and local_kw["flags"] is not None:
        local_kw["flags"] = np.asarray(local_kw["flags"], np.float32)[s_lo:s_hi]
    if local_kw.get("eps") is not None:
        local_kw["eps"] = np.asarray(local_kw["eps"], np.float32)[k_lo:k_hi]
    if k_hi > k_lo:
        local = score_fn(x[s_lo:s_hi], W, vae_lag=vae_lag, flag_mode=flag_mode, **local_kw)
    else:
        local = None
    out = {}
    want = kw.get("want", ("err", "q", "action", "margin"))
    keys = [k for k in ("err", "q", "action", "margin") if k in want]
    # shapes of an empty shard come from a peer: every rank agrees on the key set via the first
    # non-empty rank's result layout, which is static given the options
    S = 2 if flag_mode == "both" else 1
    for key in keys:
        if local is not None and key not in local:
            continue
        if local is None:
            shape = {"err": (0,), "margin": (0,), "action": (0,), "q": (0, S, 2) if S == 2 else (0, 2)}[key]
            dt = np.int32 if key == "action" else np.float32
            arr = np.zeros(shape, dt)
        else:
            arr = local[key]
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if device is not None:
            t = t.to(device)
        parts = allgather_varlen(t, counts, group)
        out[key] = torch.cat(parts, dim=0).cpu().numpy()
    return out
