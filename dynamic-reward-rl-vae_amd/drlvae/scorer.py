"""WindowScorer: the host-side handle on the MI355X scoring path (one per GPU / process).

Everything numeric happens in libdrlvae.so (hand-written gfx950 kernels); this module only
marshals numpy / torch buffers across the C-ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence

import numpy as np

from . import _abi

_MODES = {"zeros": _abi.FLAG_ZEROS, "quirk": _abi.FLAG_QUIRK, "explicit": _abi.FLAG_EXPLICIT,
          "both": _abi.FLAG_BOTH}
_RULES = {"argmax": _abi.ACTION_ARGMAX, "wadi": _abi.ACTION_WADI}


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a) -> Optional[int]:
    return None if a is None else a.ctypes.data


class WindowScorer:
    def __init__(self, device: int = 0):
        self._lib = _abi.load()
        h = C.c_void_p()
        rc = self._lib.drl_create(int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"drl_create({device}) failed ({rc}): "
                               f"{self._lib.drl_last_error(None).decode()}")
        self._h = h
        self.device = int(device)
        self.latent = 0
        self.n_actions = 2
        self.vae_dim = 0
        self.lstm_in = 0

    # ------------------------------------------------------------------ plumbing
    def close(self):
        if getattr(self, "_h", None):
            self._lib.drl_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.drl_last_error(self._h).decode()}")

    def synchronize(self):
        self._check(self._lib.drl_synchronize(self._h), "drl_synchronize")

    def set_stream(self, hip_stream: Optional[int]):
        self._check(self._lib.drl_set_stream(self._h, C.c_void_p(hip_stream or 0)), "drl_set_stream")

    def set_lstm_math(self, mode):
        """"fp32" (exact fp32 MFMA, default) or "f16x3" (3-term compensated f16 split, H=128 univariate)."""
        m = {"fp32": 0, "f16x3": 1}.get(mode, mode)
        self._check(self._lib.drl_set_lstm_math(self._h, int(m)), "drl_set_lstm_math")

    def set_knob(self, name: str, value: int):
        """A test / measurement knob of the context (include/drlvae.h: drl_set_knob)."""
        self._check(self._lib.drl_set_knob(self._h, name.encode(), int(value)), "drl_set_knob")

    def knob(self, name: str, value: int, restore: int = 0):
        """Context manager: the knob holds ``value`` inside the block and ``restore`` after it."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.set_knob(name, value)
            try:
                yield self
            finally:
                self.set_knob(name, restore)
        return cm()

    @property
    def version(self) -> str:
        return self._lib.drl_version().decode()

    def alloc_count(self) -> int:
        """Resources the library has asked the HIP runtime for since it was loaded (process-wide)."""
        return int(self._lib.drl_debug_alloc_count())

    def set_rollout_mode(self, mode):
        """"auto" (default), "fixed_point" or "sequential": which exact design ``rollout`` runs."""
        m = {"auto": 0, "fixed_point": 1, "sequential": 2}.get(mode, mode)
        self._check(self._lib.drl_set_rollout_mode(self._h, int(m)), "drl_set_rollout_mode")

    # ------------------------------------------------------------------ weights
    def set_vae(self, vae: Dict):
        """vae: {"kernels", "biases", "acts", "mu_layer", "logvar_layer", "logvar_clip"} in the
        layer order encoder hidden..., mu, logvar, decoder... (drlvae.formats.load_keras_vae /
        drlvae.compat.vae_from_dense_chain produce it)."""
        ks = [_f32(k) for k in vae["kernels"]]
        bs = [_f32(b) for b in vae["biases"]]
        n = len(ks)
        ind = (C.c_int32 * n)(*[k.shape[0] for k in ks])
        outd = (C.c_int32 * n)(*[k.shape[1] for k in ks])
        acts = (C.c_int32 * n)(*[int(a) for a in vae["acts"]])
        kp = (C.c_void_p * n)(*[k.ctypes.data for k in ks])
        bp = (C.c_void_p * n)(*[b.ctypes.data for b in bs])
        clip = vae.get("logvar_clip", 10.0)
        clip = float("inf") if clip is None else float(clip)
        lv = vae.get("logvar_layer", -1)
        self._check(self._lib.drl_set_vae(self._h, n, ind, outd, acts, int(vae["mu_layer"]),
                                          -1 if lv is None else int(lv), kp, bp, clip), "drl_set_vae")
        self.latent = ks[int(vae["mu_layer"])].shape[1]
        self.vae_dim = ks[0].shape[0]

    def set_lstm(self, lstm: Dict):
        """lstm: {"kernel" [I+H,4H] (i|j|f|o), "bias" [4H], "w_out" [H,A], "b_out" [A],
        optional "forget_bias"} (drlvae.formats.load_q_estimator produces it)."""
        k, b = _f32(lstm["kernel"]), _f32(lstm["bias"])
        wo, bo = _f32(lstm["w_out"]), _f32(lstm["b_out"])
        H = k.shape[1] // 4
        I = k.shape[0] - H
        if k.shape[1] != 4 * H or b.shape != (4 * H,) or wo.shape[0] != H or bo.shape != (wo.shape[1],):
            raise ValueError("inconsistent LSTM weight shapes")
        self._check(self._lib.drl_set_lstm(self._h, I, H, wo.shape[1], float(lstm.get("forget_bias", 1.0)),
                                           k.ctypes.data, b.ctypes.data, wo.ctypes.data, bo.ctypes.data),
                    "drl_set_lstm")
        self.n_actions = wo.shape[1]
        self.lstm_in = I

    # ------------------------------------------------------------------ options
    def _opts(self, keep, vae_lag, flag_mode, action_rule, flags=None, eps=None, in_scale=None,
              in_offset=None, vae_center=None, vae_scale=None, vae_clip=0.0, dev=False):
        o = _abi.ScoreOpts()
        o.struct_size = C.sizeof(_abi.ScoreOpts)
        o.abi_revision = _abi.ABI_REVISION
        o.vae_lag = int(vae_lag)
        o.flag_mode = _MODES[flag_mode] if isinstance(flag_mode, str) else int(flag_mode)
        o.action_rule = _RULES[action_rule] if isinstance(action_rule, str) else int(action_rule)

        def host(a):
            if a is None:
                return None
            a = _f32(a)
            keep.append(a)
            return a.ctypes.data

        if dev:
            o.flags = None if flags is None else int(flags.data_ptr())
            o.eps = None if eps is None else int(eps.data_ptr())
            keep.extend([flags, eps])
        else:
            o.flags = host(flags)
            o.eps = host(eps)
        o.in_scale = host(in_scale)
        o.in_offset = host(in_offset)
        def host64(a):
            # RobustScaler parameters cross the ABI in float64, as sklearn holds them (myasp-wadi.py:96)
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data

        o.vae_center = host64(vae_center)
        o.vae_scale = host64(vae_scale)
        o.vae_clip = float(vae_clip or 0.0)
        return o

    @staticmethod
    def window_counts(offsets: np.ndarray, W: int) -> np.ndarray:
        n = np.diff(offsets)
        return np.maximum(n - W, 0)

    # ------------------------------------------------------------------ fused path, host buffers
    def score(self, series, W: int, offsets: Optional[Sequence[int]] = None, *, vae_lag: int = 1,
              flag_mode="quirk", action_rule="argmax", want=("err", "q", "action", "margin"),
              **opt) -> Dict[str, np.ndarray]:
        """Scores every window of one series ([N] or [N,F]) or of several concatenated series
        (``offsets`` = n_seg+1 sample offsets).  Returns numpy arrays keyed by ``want``."""
        x = _f32(series)
        if x.ndim == 1:
            x = x[:, None]
        N, F = x.shape
        off = np.ascontiguousarray([0, N] if offsets is None else offsets, dtype=np.int64)
        if off[0] != 0 or off[-1] != N:
            raise ValueError("offsets must start at 0 and end at the number of samples")
        K = int(self.window_counts(off, W).sum())
        S = 2 if flag_mode in ("both", _abi.FLAG_BOTH) else 1
        keep = []
        o = self._opts(keep, vae_lag, flag_mode, action_rule, **opt)
        out = {}
        if "err" in want:
            out["err"] = np.empty(K, np.float32)
        need_q = any(w in want for w in ("q", "action", "margin"))
        if need_q:
            out["q"] = np.empty((K, S, self.n_actions) if S == 2 else (K, self.n_actions), np.float32)
        if "action" in want:
            out["action"] = np.empty(K, np.int32)
        if "margin" in want:
            out["margin"] = np.empty(K, np.float32)
        rc = self._lib.drl_score_batch(self._h, x.ctypes.data, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                       len(off) - 1, F, int(W), C.byref(o),
                                       _ptr(out.get("err")), _ptr(out.get("q")),
                                       _ptr(out.get("action")), _ptr(out.get("margin")))
        self._check(rc, "drl_score_batch")
        if need_q and "q" not in want:
            del out["q"]
        return out

    # ------------------------------------------------------------------ fused path, device buffers
    def score_dev(self, series, W: int, offsets, out: Dict, *, vae_lag: int = 1, flag_mode="zeros",
                  action_rule="argmax", noalloc: bool = False, **opt):
        """Device-resident variant: ``series`` and the tensors in ``out`` ("err", "q", "action",
        "margin"; any subset) are torch tensors on this scorer's GPU.  Asynchronous on the
        scorer's stream (see set_stream).  ``noalloc=True`` takes the entry point that allocates
        nothing (drl_score_batch_dev_noalloc: bounded sizes, see include/drlvae.h)."""
        if series.dtype.__str__() != "torch.float32" or not series.is_contiguous():
            raise ValueError("series must be a contiguous float32 tensor")
        F = 1 if series.dim() == 1 else series.shape[1]
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        keep = []
        o = self._opts(keep, vae_lag, flag_mode, action_rule, dev=True, **opt)

        def dp(name):
            t = out.get(name)
            return None if t is None else int(t.data_ptr())

        fn = self._lib.drl_score_batch_dev_noalloc if noalloc else self._lib.drl_score_batch_dev
        rc = fn(self._h, int(series.data_ptr()), off.ctypes.data_as(C.POINTER(C.c_int64)), len(off) - 1, F, int(W),
                C.byref(o), dp("err"), dp("q"), dp("action"), dp("margin"))
        self._check(rc, "drl_score_batch_dev_noalloc" if noalloc else "drl_score_batch_dev")

    # ------------------------------------------------------------------ call-surface compatibility
    def vae_predict(self, x, eps=None, return_err: bool = False):
        x = _f32(x)
        if x.ndim != 2 or x.shape[1] != self.vae_dim:
            raise ValueError(f"expected (B, {self.vae_dim}) input, got {x.shape}")
        rec = np.empty_like(x)
        err = np.empty(x.shape[0], np.float32)
        e = None if eps is None else _f32(eps)
        self._check(self._lib.drl_vae_predict(self._h, x.ctypes.data, x.shape[0], _ptr(e),
                                              rec.ctypes.data, err.ctypes.data), "drl_vae_predict")
        return (rec, err) if return_err else rec

    def q_predict(self, states):
        s = _f32(states)
        if s.ndim != 3 or s.shape[2] != self.lstm_in:
            raise ValueError(f"expected (B, W, {self.lstm_in}) states, got {s.shape}")
        q = np.empty((s.shape[0], self.n_actions), np.float32)
        self._check(self._lib.drl_q_predict(self._h, s.ctypes.data, s.shape[0], s.shape[1], q.ctypes.data),
                    "drl_q_predict")
        return q

    def rollout(self, series, W: int, offsets: Optional[Sequence[int]] = None, in_scale=None, in_offset=None,
                return_q: bool = False):
        """Greedy validation rollout with action feedback (reference myasp-A1.py:471-496) of one
        or several univariate series.  Returns actions [K] (and Q [K, A], and the pass count)."""
        x = _f32(series).reshape(-1)
        off = np.ascontiguousarray([0, len(x)] if offsets is None else offsets, dtype=np.int64)
        K = int(self.window_counts(off, W).sum())
        act = np.empty(K, np.int32)
        q = np.empty((K, self.n_actions), np.float32) if return_q else None
        sc = None if in_scale is None else _f32(in_scale)
        of = None if in_offset is None else _f32(in_offset)
        passes = C.c_int32(0)
        rc = self._lib.drl_rollout_batch(self._h, x.ctypes.data, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                         len(off) - 1, int(W), _ptr(sc), _ptr(of), act.ctypes.data, _ptr(q),
                                         C.byref(passes))
        self._check(rc, "drl_rollout_batch")
        self.last_rollout_passes = passes.value
        return (act, q) if return_q else act

    def al_get_samples(self, series, W: int, N: int, already_selected=(), offsets=None, *, flag_mode="quirk",
                       action_rule="argmax", return_margin: bool = False, **opt):
        """The N most confused windows (smallest |q0-q1|) not yet in ``already_selected``, ranked on
        the device (reference myasp-A1.py:250-259)."""
        x = _f32(series)
        if x.ndim == 1:
            x = x[:, None]
        n, F = x.shape
        off = np.ascontiguousarray([0, n] if offsets is None else offsets, dtype=np.int64)
        K = int(self.window_counts(off, W).sum())
        keep = []
        o = self._opts(keep, 1, flag_mode, action_rule, **opt)
        sel = np.ascontiguousarray(list(already_selected), dtype=np.int32)
        out = np.empty(max(int(N), 1), np.int32)
        n_out = C.c_int32(0)
        margin = np.empty(K, np.float32) if return_margin else None
        rc = self._lib.drl_al_get_samples(self._h, x.ctypes.data, off.ctypes.data_as(C.POINTER(C.c_int64)),
                                          len(off) - 1, F, int(W), C.byref(o), _ptr(sel) if len(sel) else None,
                                          len(sel), int(N), out.ctypes.data, C.byref(n_out), _ptr(margin))
        self._check(rc, "drl_al_get_samples")
        idx = [int(i) for i in out[: n_out.value]]
        return (idx, margin) if return_margin else idx

    # ------------------------------------------------------------------ measurement
    def profile(self, on: bool):
        self._check(self._lib.drl_profile_enable(self._h, 1 if on else 0), "drl_profile_enable")

    def profile_reset(self):
        self._check(self._lib.drl_profile_reset(self._h), "drl_profile_reset")

    def profile_read(self, which: int):
        ms, n = C.c_double(), C.c_int64()
        self._check(self._lib.drl_profile_read(self._h, which, C.byref(ms), C.byref(n)), "drl_profile_read")
        return ms.value, n.value

    # ------------------------------------------------------------------ the collective (RCCL)
    def comm_unique_id(self) -> bytes:
        """Rank 0 draws the communicator id; hand it to the other ranks (e.g. torch.distributed broadcast)."""
        buf = C.create_string_buffer(_abi.COMM_ID_BYTES)
        self._check(self._lib.drl_comm_unique_id(self._h, buf), "drl_comm_unique_id")
        return buf.raw

    def comm_init(self, rank: int, world: int, comm_id: bytes):
        if len(comm_id) != _abi.COMM_ID_BYTES:
            raise ValueError(f"the communicator id has {_abi.COMM_ID_BYTES} bytes")
        buf = C.create_string_buffer(bytes(comm_id), _abi.COMM_ID_BYTES)
        self._check(self._lib.drl_comm_init(self._h, int(rank), int(world), buf), "drl_comm_init")

    def allgather(self, send, recv):
        """All-gather of one contiguous device buffer per rank (torch tensors on this scorer's GPU):
        ``recv`` holds world x ``send``; asynchronous on the scorer's stream."""
        nbytes = send.numel() * send.element_size()
        if not (send.is_contiguous() and recv.is_contiguous()) or recv.numel() * recv.element_size() % max(nbytes, 1):
            raise ValueError("allgather needs contiguous buffers, recv a whole multiple of send")
        self._check(self._lib.drl_allgather_scores(self._h, int(send.data_ptr()), int(recv.data_ptr()), nbytes),
                    "drl_allgather_scores")

    def comm_size(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        n = C.c_int32(0)
        self._check(self._lib.drl_comm_size(self._h, C.byref(n)), "drl_comm_size")
        return int(n.value)

    def comm_destroy(self):
        self._check(self._lib.drl_comm_destroy(self._h), "drl_comm_destroy")

    @property
    def lstm_variant(self) -> str:
        return self._lib.drl_last_lstm_variant(self._h).decode()

    @property
    def lstm_kernels(self):
        """The LSTM kernels the last fused scoring call launched, as rocprofv3 names them."""
        k = self._lib.drl_last_lstm_kernels(self._h).decode()
        return k.split(";") if k else []

    @property
    def vae_variant(self) -> str:
        return self._lib.drl_last_vae_variant(self._h).decode()

    @property
    def lstm_tile(self) -> int:
        """Windows per workgroup tile of the last "h128i2" launch (16, 32 or 64), 0 otherwise."""
        return int(self._lib.drl_last_lstm_tile(self._h))

    @property
    def overlapped(self) -> bool:
        """True when the last scoring call ran its VAE kernel concurrently with the LSTM kernels."""
        return bool(self._lib.drl_last_call_overlapped(self._h))
