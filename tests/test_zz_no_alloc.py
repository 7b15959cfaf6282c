"""The entry point that allocates nothing (drl_score_batch_dev_noalloc, include/drlvae.h): called through
the C-ABI and checked against drl_score_batch_dev — same kernels, so bit for bit — on two calls in a row,
the second with a smaller input; both the projection paths (whose per-sample projections live in the
library's file-scope scratch array) and the univariate path; its size limits; and that it really allocates
nothing (the library's own count of its allocation sites, and hipMemGetInfo, across calls with new layouts)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W = 25


def _nets(rng, F, H):
    from test_gpu_parity import synth_lstm, synth_vae
    D = W * F
    vae = synth_vae(rng, [D, 128, 10, 128, D] if F > 1 else [D, 64, 64, 64, 10, 64, D], 1 if F > 1 else 3)
    return vae, synth_lstm(rng, F + 1, H)


def _both_ways(sc, x, offsets, kw):
    import torch
    dev = torch.device("cuda", sc.device)
    series = torch.from_numpy(x).to(dev)
    K = int(np.maximum(np.diff(offsets) - W, 0).sum())
    S = 2 if kw.get("flag_mode") == "both" else 1
    res = []
    for noalloc in (False, True):
        out = {"err": torch.full((K,), -1.0, dtype=torch.float32, device=dev),
               "q": torch.full((K, S, 2), -1.0, dtype=torch.float32, device=dev),
               "action": torch.full((K,), -1, dtype=torch.int32, device=dev),
               "margin": torch.full((K,), -1.0, dtype=torch.float32, device=dev)}
        torch.cuda.synchronize()
        sc.score_dev(series if x.shape[1] > 1 else series.reshape(-1), W, offsets, out, noalloc=noalloc, **kw)
        sc.synchronize()
        res.append({k: v.cpu().numpy() for k, v in out.items()})
    return res


@pytest.mark.parametrize("F,H", [(38, 64), (12, 128), (1, 128)])
def test_noalloc_entry_point_matches_the_allocating_one(F, H):
    from drlvae.scorer import WindowScorer
    from oracle import ref_path as orc
    rng = np.random.default_rng(1000 * F + H)
    vae, lstm = _nets(rng, F, H)
    sc = WindowScorer(0)
    try:
        sc.set_vae(vae)
        sc.set_lstm(lstm)
        D = W * F
        if F > 1:
            kw = dict(vae_lag=0, flag_mode="both", action_rule="wadi", vae_center=rng.uniform(0.3, 0.7, D),
                      vae_scale=rng.uniform(0.2, 0.6, D), vae_clip=10.0)
        else:
            kw = dict(vae_lag=1, flag_mode="quirk", action_rule="argmax", in_scale=None, in_offset=None)
        # two calls in a row on the same scratch array, the second smaller (ragged: three series, one too short)
        first = None
        for lens in ([2600, 10, 1400], [700, 333]):
            x = rng.random((sum(lens), F), dtype=np.float32)
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            ref, got = _both_ways(sc, x, off, kw)
            for k in ref:
                assert np.array_equal(ref[k].view(np.uint32), got[k].view(np.uint32)), (lens, k)
            assert np.all(got["err"] >= 0) and np.all(got["action"] >= 0)      # every output written
            if first is None:
                first = (x, off, got)
        # and the first call against the oracle, to the path's tolerance (the reference check of section 3)
        x, off, got = first
        s0 = x[:off[1]]
        if F > 1:
            want = orc.score_series_multivariate(s0[:300], W, vae, lstm, kw["vae_center"], kw["vae_scale"], 10.0,
                                                 dtype=np.float64)
        else:
            want = orc.score_series_univariate(s0[:300, 0], W, vae, lstm, "quirk", dtype=np.float64)
        n = 300 - W
        np.testing.assert_allclose(got["err"][:n], want["err"], rtol=1e-5)
        qg, qw = got["q"][:n].reshape(n, -1).astype(np.float64), np.asarray(want["q"], np.float64).reshape(n, -1)
        assert (np.abs(qg - qw).max(axis=1) / np.abs(qw).max(axis=1)).max() <= 1e-5
    finally:
        sc.close()


def test_noalloc_entry_point_rejects_sizes_beyond_its_scratch():
    import torch
    from drlvae.scorer import WindowScorer
    rng = np.random.default_rng(3)
    F, H = 38, 64
    vae, lstm = _nets(rng, F, H)
    sc = WindowScorer(0)
    try:
        sc.set_lstm(lstm)
        dev = torch.device("cuda", 0)
        n = 32_513                                      # one sample more than the projections have room for
        series = torch.zeros((n, F), dtype=torch.float32, device=dev)
        out = {"q": torch.empty((n - W, 2, 2), dtype=torch.float32, device=dev)}
        with pytest.raises(RuntimeError, match="scratch array holds"):
            sc.score_dev(series, W, np.array([0, n], np.int64), out, flag_mode="both", noalloc=True)
        sc.score_dev(series[:32_512], W, np.array([0, 32_512], np.int64), out, flag_mode="both", noalloc=True)
        sc.synchronize()
        with pytest.raises(RuntimeError, match="limit is 1024"):
            sc.score_dev(series, W, np.arange(1026, dtype=np.int64) * 30, out, flag_mode="both", noalloc=True)
    finally:
        sc.close()


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("DRLVAE_SWEEP_SEEDS", 3))))
def test_noalloc_calls_back_to_back_without_synchronising(seed):
    """Several calls of the no-allocation entry point queued on one stream with NO synchronisation in between,
    each with its own ragged layout, scaler tables and outputs: every call re-uses the one pinned staging block
    and the one scratch array (tables, then projections) of the call before, in stream order.  Results are those
    of the allocating entry point, bit for bit."""
    import torch
    from drlvae.scorer import WindowScorer
    rng = np.random.default_rng(4000 + seed)
    F = int(rng.choice([1, 5, 38]))
    H = 128 if F == 1 else int(rng.choice([64, 128]))
    vae, lstm = _nets(rng, F, H)
    sc = WindowScorer(0)
    try:
        sc.set_vae(vae)
        sc.set_lstm(lstm)
        dev = torch.device("cuda", sc.device)
        D = W * F
        calls = []
        cap = 16_000 if H > 64 else 32_000                    # samples the scratch array has projections for
        for _ in range(int(rng.integers(3, 7))):
            n_seg = int(rng.integers(1, 6))
            lens = [int(rng.integers(1, cap // n_seg)) for _ in range(n_seg)]
            lens[int(rng.integers(0, n_seg))] = max(lens[0], W + int(rng.integers(1, 500)))
            lens = [min(n, cap // n_seg) for n in lens]
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            x = rng.random((int(off[-1]), F), dtype=np.float32)
            if F > 1:
                kw = dict(vae_lag=0, flag_mode="both", action_rule="wadi", vae_center=rng.uniform(0.3, 0.7, D),
                          vae_scale=rng.uniform(0.2, 0.6, D), vae_clip=10.0)
            else:
                kw = dict(vae_lag=1, flag_mode=["quirk", "zeros"][int(rng.integers(0, 2))], action_rule="argmax",
                          in_scale=rng.uniform(0.5, 1.5, n_seg).astype(np.float32),
                          in_offset=rng.uniform(-0.1, 0.1, n_seg).astype(np.float32))
            K = int(np.maximum(np.diff(off) - W, 0).sum())
            S = 2 if kw["flag_mode"] == "both" else 1
            series = torch.from_numpy(x).to(dev)
            series = series if F > 1 else series.reshape(-1)
            outs = [{"err": torch.full((K,), -1.0, dtype=torch.float32, device=dev),
                     "q": torch.full((K, S, 2), -1.0, dtype=torch.float32, device=dev),
                     "action": torch.full((K,), -1, dtype=torch.int32, device=dev),
                     "margin": torch.full((K,), -1.0, dtype=torch.float32, device=dev)} for _ in range(2)]
            calls.append((series, off, kw, outs, K))
        torch.cuda.synchronize()
        for series, off, kw, outs, K in calls:               # queued back to back
            sc.score_dev(series, W, off, outs[1], noalloc=True, **kw)
        sc.synchronize()
        for series, off, kw, outs, K in calls:
            sc.score_dev(series, W, off, outs[0], noalloc=False, **kw)
        sc.synchronize()
        for i, (series, off, kw, outs, K) in enumerate(calls):
            for k in outs[0]:
                a, b = outs[0][k].cpu().numpy(), outs[1][k].cpu().numpy()
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (i, k, F, H, off.tolist())
            if K:
                assert float(outs[1]["err"].min()) >= 0
    finally:
        sc.close()


@pytest.mark.parametrize("F,H", [(38, 64), (12, 128), (1, 128)])
def test_noalloc_entry_point_allocates_nothing(F, H):
    """drl_debug_alloc_count (bumped at every hipMalloc / hipHostMalloc / event / stream / communicator site of
    the library) and the free device memory (hipMemGetInfo) do not move across calls of the no-allocation entry
    point — calls with layouts, scaler tables and sizes the context has never seen, which make the allocating
    entry point grow its tables and projections (shown on a second context: the counter does count)."""
    import torch
    from drlvae.scorer import WindowScorer
    rng = np.random.default_rng(77 * F + H)
    vae, lstm = _nets(rng, F, H)
    dev = torch.device("cuda", 0)
    D = W * F
    if F > 1:
        kw = dict(vae_lag=0, flag_mode="both", action_rule="wadi", vae_center=rng.uniform(0.3, 0.7, D),
                  vae_scale=rng.uniform(0.2, 0.6, D), vae_clip=10.0)
    else:
        kw = dict(vae_lag=1, flag_mode="quirk", action_rule="argmax")
    plans = [[900, 40, 300], [5000], [W + 1, 7, 2200, 1300], [12_000]]
    data = []
    for lens in plans:                                    # every tensor exists before the counters are read
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        K = int(np.maximum(np.diff(off) - W, 0).sum())
        S = 2 if F > 1 else 1
        x = torch.from_numpy(rng.random((int(off[-1]), F), dtype=np.float32)).to(dev)
        out = {"err": torch.empty(K, dtype=torch.float32, device=dev), "q": torch.empty((K, S, 2), dtype=torch.float32, device=dev),
               "action": torch.empty(K, dtype=torch.int32, device=dev), "margin": torch.empty(K, dtype=torch.float32, device=dev)}
        data.append((x if F > 1 else x.reshape(-1), off, out))
    sc = WindowScorer(0)
    other = WindowScorer(0)
    try:
        for s_ in (sc, other):
            s_.set_vae(vae)
            s_.set_lstm(lstm)
        # once through the no-allocation path: the runtime loads each kernel's code on its first launch
        sc.score_dev(data[0][0], W, data[0][1], data[0][2], noalloc=True, **kw)
        sc.synchronize()
        torch.cuda.synchronize()
        count0, free0 = sc.alloc_count(), torch.cuda.mem_get_info(0)[0]
        for x, off, out in data[1:] + data[:1]:
            sc.score_dev(x, W, off, out, noalloc=True, **kw)
        sc.synchronize()
        torch.cuda.synchronize()
        assert sc.alloc_count() == count0, "drl_score_batch_dev_noalloc reached an allocation site"
        assert torch.cuda.mem_get_info(0)[0] == free0
        # the counter is live: the allocating entry point on a fresh context does move it
        other.score_dev(data[3][0], W, data[3][1], data[3][2], **kw)
        other.synchronize()
        assert other.alloc_count() > count0
    finally:
        sc.close()
        other.close()
