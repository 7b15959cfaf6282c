"""Multi-GPU form of the scoring path: one process per GPU, weights replicated, windows sharded.

Windows are independent units, so there is no data-path exchange; the only collective is the
all-gather of the per-window scores (16-24 B per window; RCCL over xGMI with backend "nccl", gloo
on CPU for the tests).  The reference is single-process; this module has no counterpart there.

Every rank keeps its scores in ONE contiguous block of 4-byte words (ScorePack), so that a step
needs exactly one fixed-size all-gather whatever outputs it carries; ``bench.py`` and
``score_stream_sharded`` share that block and its gather.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

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


class ScorePack:
    """The score records of one rank as one block of 4-byte words:

        err[K_cap] | q[K_cap * S * A] | action[K_cap] (int32 bit patterns) | margin[K_cap]

    ``K_cap`` is the capacity every rank agrees on (the largest shard), so all blocks have the same
    size and ONE ``all_gather_into_tensor`` moves every output of a step; a rank with K < K_cap
    windows uses the head of each field.  ``device`` is a torch device (cuda for RCCL, cpu for gloo).
    """

    FIELDS = ("err", "q", "action", "margin")

    def __init__(self, K_cap: int, S: int = 1, A: int = 2, device=None):
        import torch
        self.K_cap, self.S, self.A = int(K_cap), int(S), int(A)
        self.words = self.K_cap * (3 + self.S * self.A)
        self.buf = torch.zeros(max(self.words, 1), dtype=torch.float32, device=device)

    def _views(self, buf, K: int) -> Dict:
        import torch
        Kc, SA = self.K_cap, self.S * self.A
        q = buf[Kc:Kc + K * SA]
        return {"err": buf[0:K],
                "q": q.view(K, self.S, self.A) if self.S > 1 else q.view(K, self.A),
                "action": buf[Kc * (1 + SA):Kc * (1 + SA) + K].view(torch.int32),
                "margin": buf[Kc * (2 + SA):Kc * (2 + SA) + K]}

    def views(self, K: Optional[int] = None) -> Dict:
        """Tensors aliasing the block, for the first K (default K_cap) windows: what a scoring call writes into."""
        return self._views(self.buf, self.K_cap if K is None else int(K))

    def new_gathered(self, world: int):
        import torch
        return torch.empty(world * max(self.words, 1), dtype=torch.float32, device=self.buf.device)

    def all_gather(self, gathered, group=None, async_op: bool = False):
        """The path's only collective: rank r's block lands at words [r*words, (r+1)*words) of ``gathered``."""
        import torch.distributed as dist
        return dist.all_gather_into_tensor(gathered, self.buf, group=group, async_op=async_op)

    def rank_views(self, gathered, rank: int, K: int) -> Dict:
        w = max(self.words, 1)
        return self._views(gathered[rank * w:(rank + 1) * w], K)


def score_stream_sharded(scorer, series: np.ndarray, W: int, *, vae_lag: int = 1,
                         flag_mode: str = "zeros", group=None, device=None, **kw) -> Dict:
    """Scores one long series across all ranks of ``group`` and returns, on every rank, the full
    per-window result (identical to a single-process call): each rank scores its halo'd slice into
    its ScorePack and ONE all-gather of the packed blocks assembles the whole.

    ``scorer`` is a ``WindowScorer`` with ``device`` a cuda device (device-resident: the slice is
    scored by ``score_dev`` straight into the block and gathered over RCCL, no host round trip), or,
    for CPU tests, a callable ``f(series_slice, W, vae_lag=..., flag_mode=..., **kw) -> {"err","q",
    "action","margin"}`` of numpy arrays with ``device`` None (gloo).
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
    kw.pop("want", None)
    counts = [shard_stream(n, W, r, world, vae_lag)[3] - shard_stream(n, W, r, world, vae_lag)[2] for r in range(world)]
    s_lo, s_hi, k_lo, k_hi = shard_stream(n, W, rank, world, vae_lag)
    K = k_hi - k_lo
    S = 2 if flag_mode == "both" else 1
    on_gpu = device is not None and torch.device(device).type == "cuda"
    A = int(getattr(scorer, "n_actions", 2)) if on_gpu else 2
    pack = ScorePack(max(counts) if counts else 0, S, A, device if on_gpu else None)
    local_kw = dict(kw)
    if local_kw.get("flags") is not None:
        local_kw["flags"] = np.asarray(local_kw["flags"], np.float32)[s_lo:s_hi]
    if local_kw.get("eps") is not None:
        local_kw["eps"] = np.asarray(local_kw["eps"], np.float32)[k_lo:k_hi]
    if K > 0:
        if on_gpu:
            sl = torch.from_numpy(np.ascontiguousarray(x[s_lo:s_hi])).to(device)
            for key in ("flags", "eps"):
                if local_kw.get(key) is not None:
                    local_kw[key] = torch.from_numpy(np.ascontiguousarray(local_kw[key])).to(device)
            torch.cuda.synchronize(device)                       # the scorer runs on its own stream
            scorer.score_dev(sl, W, np.array([0, s_hi - s_lo], np.int64), pack.views(K), vae_lag=vae_lag,
                             flag_mode=flag_mode, **local_kw)
            scorer.synchronize()
        else:
            res = scorer(x[s_lo:s_hi], W, vae_lag=vae_lag, flag_mode=flag_mode, **local_kw)
            v = pack.views(K)
            for key in ScorePack.FIELDS:
                v[key].copy_(torch.from_numpy(np.ascontiguousarray(res[key])).reshape(v[key].shape))
    gathered = pack.new_gathered(world)
    pack.all_gather(gathered, group)
    parts = [pack.rank_views(gathered, r, counts[r]) for r in range(world)]
    return {key: torch.cat([p[key] for p in parts], dim=0).cpu().numpy() for key in ScorePack.FIELDS}
