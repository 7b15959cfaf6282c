"""Randomised shape sweep of the HIP path against the oracle (through the C-ABI): window lengths,
feature counts, hidden widths, VAE depths, ragged batches with too-short series, every flag mode,
device-side MinMax, RobustScaler/clip — the corners the named configurations do not visit.
Same tolerance as tests/test_gpu_parity.py (1e-5 relative, fp32)."""
import os

import numpy as np
import pytest

from oracle import ref_path as orc
from test_gpu_parity import RTOL, synth_lstm, synth_vae

pytestmark = pytest.mark.gpu
# DRLVAE_SWEEP_SEEDS=N widens the two randomised sweeps to N seeds each (a one-off soak run; the default is what
# the suite runs every time)
N_UNI = int(os.environ.get("DRLVAE_SWEEP_SEEDS", 16))
N_MULTI = int(os.environ.get("DRLVAE_SWEEP_SEEDS", 12))
VAE_S_DEFAULT = 1          # the library's default of the vae_s knob


@pytest.fixture(scope="module")
def scorer():
    from drlvae.scorer import WindowScorer
    s = WindowScorer(0)
    yield s
    s.close()


def _nrel(got, want):
    """norm-wise relative error per window (rows)"""
    got = np.asarray(got, np.float64).reshape(len(got), -1)
    want = np.asarray(want, np.float64).reshape(len(want), -1)
    return (np.abs(got - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-30)).max() if len(want) else 0.0


SOAK = "DRLVAE_SWEEP_SEEDS" in os.environ


def _q_ok(got, q64, lstm):
    """The bar for Q against the fp64 oracle is 1e-5 of the window's largest |Q| (north_star), and that is what the
    suite's fixed seeds are held to.  The soak run (DRLVAE_SWEEP_SEEDS=300) meets, about once in 150 random
    networks, one whose Q head cancels to a percent of its terms or less (sum |w_out| = 81 against |Q| = 0.01);
    no float32 evaluation holds 1e-5 of |Q| there (the numpy float32 oracle is off the float64 one by 3e-6 .. 7e-6
    of |Q| on those, the kernels — v_exp_f32 / v_rcp_f32, pre-scaled gate weights — by three times that: 4e-7
    absolute on terms of magnitude 30).  The soak run therefore also accepts an absolute error of two float32 ulps
    of the head's term scale, 2 * 2^-23 * (sum_u |w_out[u, a]| + |b_out[a]|) (|h| <= 1); for a well-conditioned
    head that is below the 1e-5 bar and changes nothing."""
    got = np.asarray(got, np.float64).reshape(len(q64), -1)
    want = np.asarray(q64, np.float64).reshape(len(q64), -1)
    if not len(want):
        return True
    tol = RTOL * np.abs(want).max(axis=1)
    if SOAK:
        scale = float((np.abs(lstm["w_out"]).sum(axis=0) + np.abs(lstm["b_out"])).max())
        tol = np.maximum(tol, 2.0 * 2.0 ** -23 * scale)
    return bool((np.abs(got - want).max(axis=1) <= tol).all())


def _err_atol(err32, err64):
    """Same for the reconstruction error: a window that the VAE reconstructs almost exactly (err 1e-6 where the
    batch's typical error is 1e-2) is a difference of nearly equal numbers.  Soak run only: absolute slack of
    eight times the float32 oracle's largest deviation in the batch."""
    return 8.0 * float(np.abs(np.asarray(err32, np.float64) - err64).max()) if SOAK and len(err64) else 0.0


def _rand_vae(rng, D):
    n_enc = int(rng.integers(1, 4))
    hid = [int(rng.choice([8, 33, 64, 100, 128])) for _ in range(n_enc)]
    lat = int(rng.choice([3, 10, 32]))
    dec = int(rng.choice([16, 64, 128]))
    return synth_vae(rng, [D] + hid + [lat, dec, D], n_enc), hid, dec


@pytest.mark.parametrize("seed", range(N_UNI))
def test_univariate_random_shapes(scorer, seed):
    rng = np.random.default_rng(1000 + seed)
    Wn = int(rng.integers(2, 61))
    H = int(rng.choice([8, 32, 64, 96, 128, 128, 5, 50, 100, 127]))
    A = int(rng.choice([2, 2, 3, 4]))
    vae, _, _ = _rand_vae(rng, Wn)
    lstm = synth_lstm(rng, 2, H, A)
    n_seg = int(rng.integers(1, 6))
    lens = [int(rng.integers(max(1, Wn - 3), Wn + 180)) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = Wn + int(rng.integers(1, 150))      # at least one series with windows
    raw = [rng.normal(5.0, 3.0, n).astype(np.float32) for n in lens]
    mode = ["zeros", "quirk", "explicit"][seed % 3]
    device_minmax = bool(seed % 2)
    mm = [orc.minmax_f32(r) for r in raw]                # (normalised, scale_, min_)
    norm = [m[0] for m in mm]
    params = [(m[1], m[2]) for m in mm]
    flags = [(rng.random(n) < 0.3).astype(np.float32) for n in lens]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    scorer.set_vae(vae)
    scorer.set_lstm(lstm)
    kw = dict(flag_mode=mode)
    if mode == "explicit":
        kw["flags"] = np.concatenate(flags)
    K = int(sum(max(0, n - Wn) for n in lens))
    latent = vae["kernels"][vae["mu_layer"]].shape[1]
    eps = rng.normal(size=(K, latent)).astype(np.float32) if seed % 4 == 1 else None     # the reference's sampling path
    if eps is not None:
        kw["eps"] = eps
    if device_minmax:
        kw["in_scale"] = np.array([p[0] for p in params], np.float32)
        kw["in_offset"] = np.array([p[1] for p in params], np.float32)
        out = scorer.score(np.concatenate(raw), Wn, off, **kw)
    else:
        out = scorer.score(np.concatenate(norm), Wn, off, **kw)
    K = int(sum(max(0, n - Wn) for n in lens))
    assert out["err"].shape == (K,) and out["q"].shape == (K, A) and out["action"].shape == (K,)
    # 65..128 hidden units: the register-stationary kernel (zero-padded to 128); up to 64: projections + recurrence
    assert scorer.lstm_variant == ("h128i2" if H > 64 else "xproj+rec")
    pos = 0
    for v, fl, n in zip(norm, flags, lens):
        k = max(0, n - Wn)
        if k == 0:
            continue
        X = orc.vae_windows_univariate(v, Wn)
        e_ = None if eps is None else eps[pos:pos + k]
        rec = orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"], vae["logvar_layer"],
                              10.0, eps=e_, dtype=np.float64)
        err = orc.recon_error(X, rec, dtype=np.float64)
        err32 = orc.recon_error(X, orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"],
                                                   vae["logvar_layer"], 10.0, eps=e_, dtype=np.float32), dtype=np.float32)
        S = orc.states_list_univariate(v, Wn)
        if mode == "zeros":
            S[:, :, 1] = 0
        elif mode == "explicit":
            # flag of row t of window k = per-sample flag of that row's sample (k+1+t)
            idx = np.arange(k)[:, None] + 1 + np.arange(Wn)[None, :]
            S[:, :, 1] = fl[idx]
        q = orc.lstm_q(S, lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], 1.0, dtype=np.float64)
        np.testing.assert_allclose(out["err"][pos:pos + k], err, rtol=RTOL, atol=_err_atol(err32, err), err_msg=f"W={Wn} H={H}")
        assert _q_ok(out["q"][pos:pos + k], q, lstm), (Wn, H, A, mode, device_minmax, scorer.lstm_variant, _nrel(out["q"][pos:pos + k], q))
        srt = np.sort(q, axis=1)
        safe = (srt[:, -1] - srt[:, -2]) > 4 * RTOL * np.abs(q).max(axis=1)
        assert np.array_equal(out["action"][pos:pos + k][safe], np.argmax(q, axis=1)[safe])
        np.testing.assert_array_equal(out["margin"][pos:pos + k],
                                      np.abs(out["q"][pos:pos + k, 0] - out["q"][pos:pos + k, 1]))
        pos += k
    assert pos == K


@pytest.mark.parametrize("seed", range(N_MULTI))
def test_multivariate_random_shapes(scorer, seed):
    rng = np.random.default_rng(2000 + seed)
    Wn = int(rng.integers(3, 31))
    F = int(rng.choice([2, 5, 17, 40]))
    H = int(rng.choice([16, 40, 64, 104, 128, 3, 37, 65, 125]))
    D = Wn * F
    vae, hid, dec = _rand_vae(rng, D)
    lstm = synth_lstm(rng, F + 1, H)
    n_seg = int(rng.integers(1, 4))
    lens = [int(rng.integers(max(1, Wn - 2), Wn + 120)) for _ in range(n_seg)]
    lens[0] = Wn + int(rng.integers(1, 120))
    xs = [rng.random((n, F), dtype=np.float32) for n in lens]
    scaled = bool(seed % 2)
    center = rng.uniform(0.3, 0.7, D).astype(np.float32) if scaled else None
    scale = rng.uniform(0.05, 0.6, D).astype(np.float32) if scaled else None
    clip = 2.0 if scaled else 0.0                       # tight enough to actually clip
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    scorer.set_vae(vae)
    scorer.set_lstm(lstm)
    extra = {}
    raw = xs
    if seed % 3 == 0:                                   # per-series MinMax applied on the device
        sc_ = rng.uniform(0.5, 2.0, n_seg).astype(np.float32)
        of_ = rng.uniform(-0.5, 0.5, n_seg).astype(np.float32)
        extra = dict(in_scale=sc_, in_offset=of_)
        xs = [(x * s_ + o_).astype(np.float32) for x, s_, o_ in zip(raw, sc_, of_)]   # two fp32 roundings
    K = int(sum(max(0, n - Wn) for n in lens))
    latent = vae["kernels"][vae["mu_layer"]].shape[1]
    eps = rng.normal(size=(K, latent)).astype(np.float32) if seed % 4 == 2 else None     # the reference's sampling path
    if eps is not None:
        extra["eps"] = eps
    out = scorer.score(np.concatenate(raw), Wn, off, vae_lag=0, flag_mode="both", action_rule="wadi",
                       vae_center=center, vae_scale=scale, vae_clip=clip, **extra)
    assert out["q"].shape == (K, 2, 2)
    assert scorer.lstm_variant == ("xproj+h128" if H > 64 else "xproj+rec")
    pos = 0
    for x, n in zip(xs, lens):
        k = max(0, n - Wn)
        if k == 0:
            continue
        e_ = None if eps is None else eps[pos:pos + k]
        want = orc.score_series_multivariate(x, Wn, vae, lstm, center, scale, clip if scaled else 10.0, dtype=np.float64, eps=e_)
        want32 = orc.score_series_multivariate(x, Wn, vae, lstm, center, scale, clip if scaled else 10.0, dtype=np.float32, eps=e_)
        np.testing.assert_allclose(out["err"][pos:pos + k], want["err"], rtol=RTOL, atol=_err_atol(want32["err"], want["err"]),
                                   err_msg=f"W={Wn} F={F} hid={hid} dec={dec} {scorer._lib.drl_last_lstm_variant(scorer._h)}")
        assert _q_ok(out["q"][pos:pos + k], want["q"], lstm), (Wn, F, H, scorer.lstm_variant, _nrel(out["q"][pos:pos + k], want["q"]))
        m = np.abs(want["q"][:, 1, 1] - want["q"][:, 0, 0])
        safe = m > 4 * RTOL * np.abs(want["q"]).max(axis=(1, 2))
        assert np.array_equal(out["action"][pos:pos + k][safe], want["action"][safe])
        pos += k
    assert pos == K


_rec2 = []


def _rec2_scorer():
    """A context created with DRLVAE_REC2=2: two-candidate calls of any size take lstm_rec2_kernel (by
    default only whole rounds of 4*CUs tiles do, i.e. calls of 16k windows and more)."""
    if not _rec2:
        from drlvae.scorer import WindowScorer
        os.environ["DRLVAE_REC2"] = "2"
        try:
            _rec2.append(WindowScorer(0))
        finally:
            del os.environ["DRLVAE_REC2"]
    return _rec2[0]


@pytest.mark.parametrize("F,H,Wn", [(5, 64, 25), (38, 40, 7), (3, 16, 1), (17, 128, 12)])
def test_two_candidates_match_constant_explicit_flags_bit_for_bit(scorer, F, H, Wn):
    """DRL_FLAG_BOTH can run both candidate states of a window in one wave (lstm_rec2_kernel: shared
    projections, shared weight fragments) or one wave per candidate; explicit per-sample flags of all
    0 / all 1 run the single-candidate kernel.  All three must agree bit for bit."""
    rng = np.random.default_rng(77 + F)
    lstm = synth_lstm(rng, F + 1, H)
    scorer.set_lstm(lstm)
    lens = [Wn + 70, Wn, Wn + 1, Wn + 33]
    x = rng.random((sum(lens), F), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    both = scorer.score(x, Wn, off, vae_lag=0, flag_mode="both", want=("q",))["q"]
    assert both.shape == (104, 2, 2)
    r2 = _rec2_scorer()
    r2.set_lstm(lstm)
    both2 = r2.score(x, Wn, off, vae_lag=0, flag_mode="both", want=("q",))["q"]
    assert np.array_equal(both2.view(np.uint32), both.view(np.uint32)), (F, H)
    for cand in (0, 1):
        fl = np.full(len(x), float(cand), np.float32)
        one = scorer.score(x, Wn, off, vae_lag=0, flag_mode="explicit", flags=fl, want=("q",))["q"]
        assert np.array_equal(one.view(np.uint32), both[:, cand].view(np.uint32)), (F, H, cand)


def test_two_candidate_call_split_between_the_kernels():
    """A call of a little more than one round of tiles (4*CUs tiles of 32 windows): whole rounds run
    lstm_rec2_kernel, the remainder one wave per candidate.  Periodic input, so every period must
    score identically whichever kernel its tile lands on."""
    from drlvae.scorer import WindowScorer
    rng = np.random.default_rng(5)
    F, H, Wn, P = 6, 64, 10, 96                       # P: a multiple of the 32-window tile
    lstm = synth_lstm(rng, F + 1, H)
    sc = WindowScorer(0)
    try:
        sc.set_lstm(lstm)
        import torch
        n_cu = torch.cuda.get_device_properties(0).multi_processor_count
        n_win = (4 * n_cu + 100) * 32                   # one whole round + 100 tiles
        period = rng.random((P, F), dtype=np.float32)
        x = np.concatenate([period] * ((n_win + Wn) // P + 1))[:n_win + Wn]
        q = sc.score(x, Wn, vae_lag=0, flag_mode="both", want=("q",))["q"]
        assert q.shape == (n_win, 2, 2)
        full = q[:(n_win // P) * P].reshape(-1, P, 2, 2)
        assert np.array_equal(full.view(np.uint32), np.broadcast_to(full[:1], full.shape).view(np.uint32))
    finally:
        sc.close()


@pytest.mark.parametrize("Wn,F,H", [(1, 1, 128), (1, 3, 64), (2, 1, 8), (1, 5, 104), (191, 1, 128), (200, 1, 128)])
def test_window_length_edges(scorer, Wn, F, H):
    """W = 1 (no recurrent product at all), W = 2, and the longest windows the staged univariate
    kernel takes (W + 65 <= 256) and the first one beyond (projection path)."""
    rng = np.random.default_rng(Wn * 1000 + F * 10 + H)
    N = Wn + 150
    x = rng.random((N, F), dtype=np.float32)
    lstm = synth_lstm(rng, F + 1, H)
    scorer.set_lstm(lstm)
    out = scorer.score(x if F > 1 else x[:, 0], Wn, vae_lag=0, flag_mode="both", action_rule="wadi", want=("q", "action"))
    S = orc.states_multivariate(x, Wn)
    want = orc.lstm_q(S.reshape(-1, Wn, F + 1), lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], 1.0,
                      dtype=np.float64).reshape(N - Wn, 2, 2)
    assert out["q"].shape == want.shape
    tol = RTOL if Wn < 100 else 4 * RTOL          # 200 recurrent steps accumulate fp32 rounding
    assert _nrel(out["q"], want) <= tol, (scorer.lstm_variant, _nrel(out["q"], want))


def test_repeated_calls_are_bit_identical():
    """A missing barrier or a buffer reused too early shows up as run-to-run differences long before
    it shows up as a wrong answer: score the same inputs many times, on every kernel family and with
    the two-stream overlap forced on, and require identical bits every time."""
    import os
    from drlvae.scorer import WindowScorer
    old = os.environ.get("DRLVAE_OVERLAP")
    os.environ["DRLVAE_OVERLAP"] = "2"                  # read at drl_create
    sc = WindowScorer(0)
    if old is None:
        del os.environ["DRLVAE_OVERLAP"]
    else:
        os.environ["DRLVAE_OVERLAP"] = old
    try:
        rng = np.random.default_rng(77)
        cases = []
        # univariate H=128: small call (32-window tiles) and large ragged call (64-window tiles, persistent)
        vae_u, _, _ = _rand_vae(rng, 25)
        lstm_u = synth_lstm(rng, 2, 128)
        lens = [int(n) for n in rng.integers(700, 1500, 40)]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        big = rng.random(int(off[-1]), dtype=np.float32)
        cases.append(("uni-small", vae_u, lstm_u, big[:1400], None, dict(flag_mode="quirk"), "fp32"))
        cases.append(("uni-large", vae_u, lstm_u, big, off, dict(flag_mode="quirk"), "fp32"))
        cases.append(("uni-large-f16x3", vae_u, lstm_u, big, off, dict(flag_mode="zeros"), "f16x3"))
        # multivariate, wide VAE, both LSTM paths
        for H in (64, 128):
            F = 38
            vae_m = synth_vae(rng, [25 * F, 128, 10, 128, 25 * F], 1)
            x = rng.random((6000, F), dtype=np.float32)
            cen = rng.uniform(0.3, 0.7, 25 * F).astype(np.float32)
            scl = rng.uniform(0.2, 0.6, 25 * F).astype(np.float32)
            cases.append((f"multi-H{H}", vae_m, synth_lstm(rng, F + 1, H), x, None,
                          dict(vae_lag=0, flag_mode="both", action_rule="wadi", vae_center=cen, vae_scale=scl,
                               vae_clip=10.0), "fp32"))
        for name, vae, lstm, x, offs, kw, math in cases:
            sc.set_vae(vae)
            sc.set_lstm(lstm)
            sc.set_lstm_math(math)
            first = sc.score(x, 25, offs, **kw)
            assert sc.overlapped
            for rep in range(25):
                again = sc.score(x, 25, offs, **kw)
                for key in first:
                    assert np.array_equal(first[key].view(np.uint32), again[key].view(np.uint32)), (name, key, rep)
    finally:
        sc.close()


@pytest.mark.parametrize("F,H", [(1, 128), (3, 64), (1, 64)])
def test_network_without_flag_channel(scorer, F, H):
    """in_dim == F: the reference's legacy estimators (AdaptiveTauRL: kernel [129, 512]) see the values only."""
    rng = np.random.default_rng(900 + F + H)
    Wn, N = 25, 400
    x = rng.random((N, F), dtype=np.float32)
    lstm = synth_lstm(rng, F, H)
    scorer.set_lstm(lstm)
    out = scorer.score(x if F > 1 else x[:, 0], Wn, vae_lag=0, flag_mode="zeros", want=("q", "action"))
    idx = np.arange(N - Wn)[:, None] + 1 + np.arange(Wn)[None, :]
    want = orc.lstm_q(x[idx], lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], 1.0, dtype=np.float64)
    assert _nrel(out["q"], want) <= RTOL
    with pytest.raises(RuntimeError, match="no flag channel"):
        scorer.score(x if F > 1 else x[:, 0], Wn, vae_lag=0, flag_mode="quirk", want=("q",))


@pytest.mark.parametrize("F,last_act", [(1, 0), (1, 1), (8, 0), (8, 1), (8, 2)])
def test_vae_output_activations(scorer, F, last_act):
    """Linear / relu / sigmoid output layers (the reference's KPI VAE ends in a linear Dense), on the
    chain kernel (D = 25) and on the wide kernel (D = 200)."""
    rng = np.random.default_rng(50 + F * 3 + last_act)
    Wn, N = 25, 500
    D = Wn * F
    vae = synth_vae(rng, [D, 128, 10, 128, D], 1)
    vae["acts"][-1] = last_act
    x = rng.random((N, F), dtype=np.float32)
    scorer.set_vae(vae)
    out = scorer.score(x if F > 1 else x[:, 0], Wn, vae_lag=0, want=("err",))
    idx = np.arange(N - Wn)[:, None] + 1 + np.arange(Wn)[None, :]
    X = x[idx].reshape(N - Wn, D)
    rec = orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"], vae["logvar_layer"], 10.0,
                          dtype=np.float64)
    np.testing.assert_allclose(out["err"], orc.recon_error(X, rec, dtype=np.float64), rtol=RTOL)


def test_many_short_series_and_a_very_wide_one(scorer):
    """20,000 series of 10..200 samples (most of the segment-table lookups leave the LDS copy: more
    than 512 series) and a 300-feature stream whose windows (D = 7,500) are too wide for the LDS-staged
    VAE kernel: spot-checked against the oracle, and against single-series calls bit for bit."""
    rng = np.random.default_rng(2024)
    vae, _, _ = _rand_vae(rng, 25)
    lstm = synth_lstm(rng, 2, 128)
    scorer.set_vae(vae)
    scorer.set_lstm(lstm)
    lens = rng.integers(10, 201, 20000)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    x = rng.random(int(off[-1]), dtype=np.float32)
    out = scorer.score(x, 25, off, flag_mode="quirk")
    kcount = np.maximum(lens - 25, 0)
    woff = np.concatenate([[0], np.cumsum(kcount)])
    assert len(out["err"]) == woff[-1]
    for i in (0, 1, 777, 5000, 12345, 19999):
        k = int(kcount[i])
        if k == 0:
            continue
        v = x[off[i]:off[i + 1]]
        one = scorer.score(v, 25, flag_mode="quirk")
        for key in out:
            assert np.array_equal(one[key].view(np.uint32), out[key][woff[i]:woff[i] + k].view(np.uint32)), (i, key)
        want = orc.score_series_univariate(v, 25, vae, lstm, "quirk", dtype=np.float64)
        np.testing.assert_allclose(one["err"], want["err"], rtol=RTOL)
        # these random heads put Q near zero (a sum of 128 terms of size ~0.5 cancelling to ~0.1), so the
        # error is measured against the series' largest |Q|, not window by window
        assert np.abs(one["q"] - want["q"]).max() / np.abs(want["q"]).max() <= RTOL
    # very wide windows
    F, N = 300, 700
    xm = rng.random((N, F), dtype=np.float32)
    vae_w = synth_vae(rng, [25 * F, 128, 10, 128, 25 * F], 1)
    lstm_w = synth_lstm(rng, F + 1, 64)
    lstm_w["kernel"][:F] *= 0.2
    scorer.set_vae(vae_w)
    scorer.set_lstm(lstm_w)
    got = scorer.score(xm, 25, vae_lag=0, flag_mode="both", action_rule="wadi")
    want = orc.score_series_multivariate(xm, 25, vae_w, lstm_w, None, None, 10.0, dtype=np.float64)
    np.testing.assert_allclose(got["err"], want["err"], rtol=RTOL)
    assert _nrel(got["q"], want["q"]) <= RTOL


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 8))))
def test_rollout_designs_agree_on_random_networks(scorer, seed):
    """drl_rollout_batch on random networks, whose Q head changes sign often (many action changes, long fixed-point
    iterations, long lists of tiles for the sparse passes): the fixed point with sparse passes, the fixed point with
    full passes only, and the sequential kernel give the same actions and the same Q bits, and the fixed point takes
    the same number of passes either way.  (Against the numpy oracle such a rollout is chaotic — one near-tie flips
    everything after it — so that comparison is made on the reference's checkpoints, test_gpu_parity.py.)"""
    rng = np.random.default_rng(9000 + seed)
    Wn = int(rng.integers(2, 61))
    H = int(rng.choice([16, 64, 104, 128, 128, 7, 77]))
    lstm = synth_lstm(rng, 2, H, 2)
    n_seg = int(rng.integers(1, 8))
    lens = [int(rng.integers(max(1, Wn - 3), Wn + 500)) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = Wn + int(rng.integers(1, 400))
    v = rng.random(sum(lens), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    scorer.set_lstm(lstm)
    try:
        scorer.set_rollout_mode("fixed_point")
        a_sp, q_sp = scorer.rollout(v, Wn, off, return_q=True)
        passes = scorer.last_rollout_passes
        with scorer.knob("rollout_dense", 1):
            a_de, q_de = scorer.rollout(v, Wn, off, return_q=True)
            assert scorer.last_rollout_passes == passes
        scorer.set_rollout_mode("sequential")
        a_sq, q_sq = scorer.rollout(v, Wn, off, return_q=True)
        scorer.set_rollout_mode("auto")
        a_au = scorer.rollout(v, Wn, off)
    finally:
        scorer.set_rollout_mode("auto")
    K = int(sum(max(0, n - Wn) for n in lens))
    assert a_sp.shape == (K,) and passes >= 1
    for a, q in ((a_de, q_de), (a_sq, q_sq)):
        assert np.array_equal(a, a_sp) and np.array_equal(q.view(np.uint32), q_sp.view(np.uint32)), (Wn, H, lens, passes)
    assert np.array_equal(a_au, a_sp)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 6))))
def test_pipelined_host_path_random_chunk_plans(scorer, seed):
    """The chunk planner of drl_score_batch's pipelined path (whole series, window ranges of long series with their
    halo, series too short for a window in between) on random ragged batches and random chunk sizes: the result is
    that of the one-shot call, bit for bit, in every output."""
    rng = np.random.default_rng(7000 + seed)
    F = int(rng.choice([1, 1, 3, 8]))
    Wn = int(rng.integers(2, 40))
    H = 128 if F == 1 and seed % 2 == 0 else 64
    mode = ["zeros", "quirk", "explicit", "both"][int(rng.integers(0, 4))] if F > 1 or True else "zeros"
    lstm = synth_lstm(rng, F + 1, H, 2)
    vae, _, _ = _rand_vae(rng, Wn * F)
    n_seg = int(rng.integers(1, 9))
    lens = [int(rng.choice([rng.integers(1, Wn + 2), rng.integers(Wn, 4000), rng.integers(4000, 30000)])) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = int(rng.integers(2000, 30000))
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    x = rng.random((int(off[-1]), F), dtype=np.float32)
    kw = dict(flag_mode=mode, vae_lag=int(rng.integers(0, 2)))
    if mode == "explicit":
        kw["flags"] = (rng.random(int(off[-1])) < 0.3).astype(np.float32)
    if mode == "both":
        kw["action_rule"] = "wadi"
    if rng.random() < 0.5:
        kw["in_scale"] = rng.uniform(0.5, 1.5, n_seg).astype(np.float32)
        kw["in_offset"] = rng.uniform(-0.2, 0.2, n_seg).astype(np.float32)
    scorer.set_vae(vae)
    scorer.set_lstm(lstm)
    with scorer.knob("no_pipeline", 1):
        ref = scorer.score(x, Wn, off, **kw)
    with scorer.knob("pipe_min", 1), scorer.knob("pipe_chunk", int(rng.choice([4 * Wn, 777, 5000, 20000, 100000]))):
        got = scorer.score(x, Wn, off, **kw)
    K = int(sum(max(0, n - Wn) for n in lens))
    assert got["err"].shape == (K,)
    for k in ref:
        assert np.array_equal(ref[k].view(np.uint32), got[k].view(np.uint32)), (k, F, Wn, H, mode, lens)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 6))))
def test_al_selection_random_batches(scorer, seed):
    """drl_al_get_samples on random ragged batches: the device selection is numpy's stable ranking of the device's
    own margins with the already-labelled windows removed, first N (reference myasp-A1.py:250-259); duplicate and
    out-of-range entries in already_selected, N larger than what is left, exact ties (a constant series gives every
    window the same margin: ties keep window order)."""
    rng = np.random.default_rng(11000 + seed)
    Wn = int(rng.integers(2, 40))
    H = int(rng.choice([64, 128, 24]))
    lstm = synth_lstm(rng, 2, H, 2)
    n_seg = int(rng.integers(1, 7))
    lens = [int(rng.integers(1, 3000)) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = Wn + int(rng.integers(1, 3000))
    v = rng.random(sum(lens), dtype=np.float32)
    if seed % 4 == 3:
        v[:] = 0.5                                            # every window identical: all margins tie
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    K = int(sum(max(0, n - Wn) for n in lens))
    n_sel = int(rng.integers(0, max(1, K // 2)))
    already = [int(i) for i in rng.integers(0, K, n_sel)]      # may repeat
    if seed % 3 == 0:
        already += [K + 5, -3]                                # out of range: ignored
    N = int(rng.choice([0, 1, 10, 1000, K + 10]))
    scorer.set_lstm(lstm)
    idx, margin = scorer.al_get_samples(v, Wn, N, already, offsets=off, flag_mode=["quirk", "zeros"][seed % 2],
                                        return_margin=True)
    assert margin.shape == (K,)
    sel = set(i for i in already if 0 <= i < K)
    want = [int(i) for i in np.argsort(margin, kind="stable") if int(i) not in sel][:N]
    assert idx == want, (Wn, H, lens, N, len(sel))


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 8))))
def test_estimator_predict_random_batches(scorer, seed):
    """drl_q_predict (estimator.predict on explicit states) on random shapes and batch sizes: a state's Q does not
    depend on the batch it arrives in — one by one (the matrix-vector kernel), as a whole batch (matrix-vector or
    MFMA kernels, by size), and through the fallback kernel — bit for bit; and it is the oracle's to 1e-5.
    drl_vae_predict likewise: rows one by one equal the batch."""
    rng = np.random.default_rng(13000 + seed)
    I = int(rng.choice([1, 2, 2, 3, 9, 39, 124]))
    H = int(rng.choice([64, 128, 128, 40, 100]))
    A = int(rng.choice([2, 2, 3]))
    Wn = int(rng.integers(1, 64))
    B = int(rng.choice([1, 2, 5, 33, 300, 2100]))
    lstm = synth_lstm(rng, I, H, A)
    lstm["kernel"][:I] *= min(1.0, 4.0 / I)                  # keep the pre-activations in the cell's active range
    scorer.set_lstm(lstm)
    S = rng.random((B, Wn, I), dtype=np.float32)
    q = scorer.q_predict(S)
    variant = scorer.lstm_variant
    want = orc.lstm_q(S, lstm["kernel"], lstm["bias"], lstm["w_out"], lstm["b_out"], 1.0, dtype=np.float64)
    assert _q_ok(q, want, lstm), (I, H, A, Wn, B, variant, _nrel(q, want))
    pick = sorted(set(int(i) for i in rng.integers(0, B, 4)))
    for i in pick:
        one = scorer.q_predict(S[i:i + 1])
        assert np.array_equal(one.view(np.uint32), q[i:i + 1].view(np.uint32)), (I, H, Wn, B, i, variant, scorer.lstm_variant)
    os.environ["DRLVAE_FORCE_GENERIC"] = "1"
    try:
        from drlvae.scorer import WindowScorer
        g = WindowScorer(0)
        try:
            g.set_lstm(lstm)
            qg = g.q_predict(S[:64])
            assert g.lstm_variant in ("generic", "gemv", "h128i2")     # the univariate H > 64 net has its own explicit-state kernel
            assert np.array_equal(qg.view(np.uint32), q[:64].view(np.uint32)), (I, H, Wn, B, variant, g.lstm_variant)
        finally:
            g.close()
    finally:
        del os.environ["DRLVAE_FORCE_GENERIC"]
    D = int(rng.integers(2, 200))
    vae, _, _ = _rand_vae(rng, D)
    scorer.set_vae(vae)
    X = rng.random((int(rng.choice([1, 3, 40, 700])), D), dtype=np.float32)
    rec, err = scorer.vae_predict(X, return_err=True)
    r64 = orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"], vae["logvar_layer"], 10.0, dtype=np.float64)
    np.testing.assert_allclose(rec, r64, rtol=RTOL, atol=2e-7)
    for i in sorted(set(int(i) for i in rng.integers(0, len(X), 3))):
        r1, e1 = scorer.vae_predict(X[i:i + 1], return_err=True)
        assert np.array_equal(r1.view(np.uint32), rec[i:i + 1].view(np.uint32)) and e1[0] == err[i], (D, len(X), i)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 8))))
def test_argmax_epilogue_in_the_lstm_kernels_equals_finalize_kernel(scorer, seed):
    """Action and margin are written by the LSTM kernel's own epilogue whenever the workgroup that computes a
    window's Q holds every value the rule needs (one candidate state; both candidates in one wave), by
    finalize_kernel from q otherwise.  Both must give the same actions and margins, bit for bit, and the margin
    is |q0 - q1| of the q that was written."""
    rng = np.random.default_rng(15000 + seed)
    F = int(rng.choice([1, 1, 4, 20]))
    H = int(rng.choice([128, 64, 30])) if F > 1 else int(rng.choice([128, 128, 64, 90]))
    A = int(rng.choice([2, 2, 3, 5])) if F == 1 else 2
    Wn = int(rng.integers(1, 50))
    lstm = synth_lstm(rng, F + 1, H, A)
    lstm["b_out"] = (rng.normal(size=A) * 0.05).astype(np.float32)
    n_seg = int(rng.integers(1, 6))
    lens = [int(rng.integers(1, 2500)) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = Wn + int(rng.integers(1, 2500))
    x = rng.random((sum(lens), F), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    if F > 1 and seed % 2 == 0:
        kw = dict(flag_mode="both", action_rule="wadi", vae_lag=0)
    elif F > 1:
        kw = dict(flag_mode="both", action_rule="argmax", vae_lag=0)
    else:
        kw = dict(flag_mode=["quirk", "zeros"][seed % 2])
    sc = _rec2_scorer() if (F > 1 and seed % 4 < 2) else scorer       # both candidates in one wave for every tile
    sc.set_lstm(lstm)
    xs = x if F > 1 else x[:, 0]
    got = sc.score(xs, Wn, off, want=("q", "action", "margin"), **kw)
    with sc.knob("no_fused_epilogue", 1):
        ref = sc.score(xs, Wn, off, want=("q", "action", "margin"), **kw)
    for k in ("q", "action", "margin"):
        assert np.array_equal(ref[k].view(np.uint32), got[k].view(np.uint32)), (k, F, H, A, Wn, kw, sc.lstm_variant)
    q = got["q"]
    if kw.get("action_rule") == "wadi":
        q0, q1 = q[:, 0, 0], q[:, 1, 1]
        assert np.array_equal(got["action"], (q1 > q0).astype(np.int32))
    else:
        qa = q[:, 0, :] if q.ndim == 3 else q
        q0, q1 = qa[:, 0], qa[:, 1]
        assert np.array_equal(got["action"], np.argmax(qa, axis=1).astype(np.int32))
    assert np.array_equal(got["margin"], np.abs(q0 - q1))


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 10))))
def test_wide_vae_kernel_variants_equal_the_one_tile_kernel(scorer, seed):
    """vae_wide16s_kernel (software-pipelined: every memory operation of a slab behind an MFMA of the slab before; the
    default where two of its workgroups fit a CU) and vae_wide16pp_kernel (two 64-window tiles per workgroup in
    ping-pong over one weight ring, persistent grid, 16-byte region loads; opt-in) perform vae_wide16_kernel's MFMAs in
    the same order per accumulator: the reconstruction error must be the same bit for bit — ragged batches (odd tile counts, tiles of fewer than 64 windows, series too
    short for a window), with and without the RobustScaler transform, device MinMax, both window lags, hidden widths
    40..128, D not a multiple of 32; and it is held to the oracle like every other kernel."""
    rng = np.random.default_rng(7000 + seed)
    Wn = int(rng.integers(4, 31))
    F = int(rng.choice([6, 17, 38, 123]))
    hid = int(rng.choice([40, 64, 97, 112, 128]))           # (up to 64: the instantiation that skips the zero row tiles)
    D = Wn * F
    if D <= 128:
        F = 38
        D = Wn * F
    n_enc = 1 + seed % 2
    vae = synth_vae(rng, [D] + [hid] * n_enc + [int(rng.choice([3, 10, 32])), hid, D], n_enc)
    n_seg = int(rng.integers(1, 6))
    lens = [int(rng.integers(max(1, Wn - 2), Wn + 400)) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = Wn + int(rng.integers(1, 700))
    x = rng.random((sum(lens), F), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    kw = dict(vae_lag=int(seed % 3 != 0), want=("err",))
    scaled = seed % 4 != 3
    if scaled:
        kw.update(vae_center=rng.uniform(0.3, 0.7, D), vae_scale=rng.uniform(0.05, 0.6, D), vae_clip=2.0)
    if seed % 2:
        kw.update(in_scale=rng.uniform(0.5, 2.0, n_seg).astype(np.float32), in_offset=rng.uniform(-0.5, 0.5, n_seg).astype(np.float32))
    scorer.set_vae(vae)
    with scorer.knob("vae_pp", 0), scorer.knob("vae_s", 0, restore=VAE_S_DEFAULT):
        ref = scorer.score(x, Wn, off, **kw)["err"]
        assert scorer.vae_variant == "wide16"
    with scorer.knob("vae_pp", 2):
        got = scorer.score(x, Wn, off, **kw)["err"]
        assert scorer.vae_variant == "wide16pp"
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), (Wn, F, hid, lens, kw.keys())
    with scorer.knob("vae_s", 1, restore=VAE_S_DEFAULT):
        got_s = scorer.score(x, Wn, off, **kw)["err"]
        assert scorer.vae_variant == "wide16s"
    assert np.array_equal(ref.view(np.uint32), got_s.view(np.uint32)), ("wide16s", Wn, F, hid, lens, kw.keys())
    # and against the oracle (first series that has windows)
    i = next(i for i, n in enumerate(lens) if n > Wn)
    xs = x[off[i]:off[i + 1]][:Wn + 120]
    if "in_scale" in kw:
        xs = (xs * kw["in_scale"][i] + kw["in_offset"][i]).astype(np.float32)
    K0 = int(np.maximum(np.diff(off)[:i] - Wn, 0).sum())
    lo = 1 - kw["vae_lag"]
    X = orc.vae_windows_multivariate(xs, Wn, kw.get("vae_center"), kw.get("vae_scale"), 2.0 if scaled else None, lo, lo + len(xs) - Wn)
    rec = orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"], vae["logvar_layer"], dtype=np.float64)
    np.testing.assert_allclose(got[K0:K0 + len(X)], orc.recon_error(X, rec, np.float64), rtol=RTOL)


@pytest.mark.parametrize("seed", range(int(os.environ.get("DRLVAE_SWEEP_SEEDS", 8))))
def test_eps_on_the_pipelined_wide_kernel_equals_the_chain_kernel(scorer, seed):
    """The reference's sampled z (myasp-A1.py:84, 94; myasp-wadi.py:106-108) on wide windows: vae_wide16s_kernel with
    the logvar head in its middle layers against the chain kernel (vae_kernel<4>) on the same draws — ragged batches,
    tiles of fewer than 64 windows, series too short for a window, latent widths 3..32 (one and two 16-row tiles),
    one or two encoder layers, logvar clip on and off — and against the fp64 oracle on the first series."""
    rng = np.random.default_rng(9100 + seed)
    Wn = int(rng.integers(5, 28))
    F = int(rng.choice([17, 38, 123]))
    hid = int(rng.choice([48, 64, 97, 120, 128]))
    if Wn * F <= 128:                                       # (narrow windows take the chain kernel whatever the hidden width)
        F = 38
    D = Wn * F
    latent = int(rng.choice([3, 10, 16, 17, 32]))
    n_enc = 1 + seed % 2
    vae = synth_vae(rng, [D] + [hid] * n_enc + [latent, hid, D], n_enc)
    n_seg = int(rng.integers(1, 5))
    lens = [int(rng.integers(max(1, Wn - 2), Wn + 300)) for _ in range(n_seg)]
    lens[int(rng.integers(0, n_seg))] = Wn + int(rng.integers(1, 500))
    x = rng.random((sum(lens), F), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    K = int(np.maximum(np.diff(off) - Wn, 0).sum())
    eps = rng.normal(size=(K, latent)).astype(np.float32)
    kw = dict(vae_lag=int(seed % 3 != 0), want=("err",), eps=eps)
    if seed % 4 != 3:
        kw.update(vae_center=rng.uniform(0.3, 0.7, D), vae_scale=rng.uniform(0.05, 0.6, D), vae_clip=2.0)
    scorer.set_vae(vae)
    got = scorer.score(x, Wn, off, **kw)["err"]
    assert scorer.vae_variant == "wide16s"
    with scorer.knob("vae_s", 0, restore=VAE_S_DEFAULT):
        ref = scorer.score(x, Wn, off, **kw)["err"]
        assert scorer.vae_variant.startswith("chain")
    # two float32 kernels with different summation orders (16x16x4 slabs of 32 against 32x32x2 tiles): a few ulps of a
    # mean of thousands of squares; the 600-seed soak run has one window in 150,000 at 2.2e-6
    np.testing.assert_allclose(got, ref, rtol=5e-6)
    i = next(i for i, n in enumerate(lens) if n > Wn)
    xs = x[off[i]:off[i + 1]][:Wn + 100]
    K0 = int(np.maximum(np.diff(off)[:i] - Wn, 0).sum())
    lo = 1 - kw["vae_lag"]
    X = orc.vae_windows_multivariate(xs, Wn, kw.get("vae_center"), kw.get("vae_scale"), 2.0 if "vae_clip" in kw else None, lo, lo + len(xs) - Wn)
    rec = orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], vae["mu_layer"], vae["logvar_layer"], 10.0,
                          eps=eps[K0:K0 + len(X)], dtype=np.float64)
    np.testing.assert_allclose(got[K0:K0 + len(X)], orc.recon_error(X, rec, np.float64), rtol=RTOL)


def test_wide_kernels_when_only_one_workgroup_fits_a_cu(scorer):
    """F = 140 sensors at W = 25: the flat region of 64 windows is 49 KB and two software-pipelined workgroups no longer
    fit a CU's LDS; vae_wide16s_kernel then runs alone on its CU (still 17 % faster than vae_wide16_kernel there:
    tools/wide_f140.py).  Same bits as vae_wide16_kernel (vae_s = 0), the oracle's error, and the same draws with eps
    through it and through the chain kernel."""
    rng = np.random.default_rng(77)
    Wn, F, hid, latent = 25, 140, 128, 10
    D = Wn * F
    vae = synth_vae(rng, [D, hid, latent, hid, D], 1)
    lens = [Wn + 150, Wn - 1, Wn + 70]
    x = rng.random((sum(lens), F), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    kw = dict(vae_lag=0, want=("err",), vae_center=rng.uniform(0.3, 0.7, D), vae_scale=rng.uniform(0.05, 0.6, D), vae_clip=2.0)
    scorer.set_vae(vae)
    got = scorer.score(x, Wn, off, **kw)["err"]
    assert scorer.vae_variant == "wide16s"
    with scorer.knob("vae_s", 0, restore=VAE_S_DEFAULT):
        other = scorer.score(x, Wn, off, **kw)["err"]
        assert scorer.vae_variant == "wide16"
    assert np.array_equal(other.view(np.uint32), got.view(np.uint32))
    X = orc.vae_windows_multivariate(x[:lens[0]], Wn, kw["vae_center"], kw["vae_scale"], 2.0, 1, 1 + lens[0] - Wn)
    rec = orc.vae_forward(X, vae["kernels"], vae["biases"], vae["acts"], 1, 2, dtype=np.float64)
    np.testing.assert_allclose(got[:len(X)], orc.recon_error(X, rec, np.float64), rtol=RTOL)
    eps = rng.normal(size=(len(got), latent)).astype(np.float32)
    e1 = scorer.score(x, Wn, off, eps=eps, **kw)["err"]
    assert scorer.vae_variant == "wide16s"
    with scorer.knob("vae_s", 0, restore=VAE_S_DEFAULT):
        e0 = scorer.score(x, Wn, off, eps=eps, **kw)["err"]
        assert scorer.vae_variant.startswith("chain")
    np.testing.assert_allclose(e1, e0, rtol=5e-6)


@pytest.mark.parametrize("mode,S", [("both", 2), ("zeros", 1)])
def test_h128p_last_round_as_half_tiles_is_bit_identical(scorer, mode, S):
    """Multivariate streams with more than 64 hidden units (lstm_h128p_kernel): when the last round of 64-window tiles
    is at most half full the library runs it as 32-window tiles in a second launch (knob h128_split).  Q, action and
    margin must be the same bits as the one-launch form (which the other tests hold to the oracle) — on a ragged batch
    whose split point falls inside a series, with series too short for a window in front of and behind it."""
    rng = np.random.default_rng(4400 + S)
    Wn, F, H = 12, 3, 96
    G = 256 // S                                            # persistent workgroups per candidate state on an MI355X
    lens = [Wn - 1, Wn + 64 * 40 + 7, Wn + 3, 0, Wn + 64 * (G - 45) + 33, Wn + 64 * 30 + 20, 5]     # G + 29 tiles of 64
    x = rng.random((sum(lens), F), dtype=np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    scorer.set_lstm(synth_lstm(rng, F + 1, H))
    kw = dict(vae_lag=0, flag_mode=mode, want=("q", "action", "margin"))
    if S == 2:
        kw["action_rule"] = "wadi"
    got = scorer.score(x, Wn, off, **kw)
    assert scorer.lstm_kernels == ["xproj_kernel<8>", "lstm_h128p_kernel<2>", "lstm_h128p_kernel<1>"], scorer.lstm_kernels
    with scorer.knob("h128_split", 0, restore=1):
        ref = scorer.score(x, Wn, off, **kw)
        assert scorer.lstm_kernels == ["xproj_kernel<8>", "lstm_h128p_kernel<2>"], scorer.lstm_kernels
    for k in ("q", "action", "margin"):
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k


def test_h128i2_last_round_as_half_tiles_is_bit_identical(scorer):
    """The univariate persistent kernel: 256 workgroups walk 64-window tiles; a last round that is at most half full runs
    as 32-window tiles in a second launch (lstm_h128i2_kernel<1>).  Same bits as the one-launch form on a ragged batch
    (split point inside a series; too-short and empty series around it), with the flag quirk and device MinMax."""
    rng = np.random.default_rng(5151)
    Wn = 25
    lens = [Wn - 3, Wn + 64 * 100 + 11, 0, Wn + 2, Wn + 64 * 170 + 40, Wn + 64 * 25 + 63, 7]      # 256 + 42 tiles of 64
    x = rng.random(sum(lens)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    n_seg = len(lens)
    scorer.set_lstm(synth_lstm(rng, 2, 128))
    kw = dict(vae_lag=1, flag_mode="quirk", want=("q", "action", "margin"),
              in_scale=rng.uniform(0.5, 2.0, n_seg).astype(np.float32), in_offset=rng.uniform(-0.5, 0.5, n_seg).astype(np.float32))
    got = scorer.score(x, Wn, off, **kw)
    assert scorer.lstm_kernels == ["lstm_h128i2_kernel<2>", "lstm_h128i2_kernel<1>"], scorer.lstm_kernels
    with scorer.knob("h128_split", 0, restore=1):
        ref = scorer.score(x, Wn, off, **kw)
        assert scorer.lstm_kernels == ["lstm_h128i2_kernel<2>"], scorer.lstm_kernels
    for k in ("q", "action", "margin"):
        assert np.array_equal(got[k].view(np.uint32), ref[k].view(np.uint32)), k
