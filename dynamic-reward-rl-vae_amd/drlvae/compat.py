"""Host-side mirror of the reference's call surface for the scored path.

A maintainer of the reference swaps the imports and keeps calling
``vae.predict(x)``, ``estimator.predict(state=[s], sess=sess)``, ``env.statefnc`` /
``env.rewardfnc``, ``penalty_array[c]`` and ``active_learning.get_samples()`` unchanged.  All
numerics run in libdrlvae.so on the GPU; what stays in Python here is what is Python glue in the
reference too (array assembly, reward constants, argsort).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np

from .formats.keras_h5 import load_keras_vae
from .formats.tf_bundle import load_q_estimator, load_vae_dense_chain
from .scorer import WindowScorer

# ------------------------------------------------------------------------------------------ a-1


def minmax_params(x, dtype=np.float32):
    """``MinMaxScaler().fit`` parameters (scale_, min_) as sklearn computes them in ``dtype``
    (reference env.py:75-79 casts to float32 first; env_smd.py:76-77 fits in float64)."""
    x = np.asarray(x, dtype=dtype)
    mn, mx = x.min(), x.max()
    rng = dtype(mx - mn)
    if rng == 0 or rng < 10 * np.finfo(dtype).eps:
        rng = dtype(1.0)
    scale = dtype(1.0) / rng
    return scale, dtype(0.0) - mn * scale


def minmax_normalise(x, dtype=np.float32) -> np.ndarray:
    """The value column as the reference's environments hand it to the path: ``MinMaxScaler().
    fit_transform`` — sklearn's ``x*scale_ + min_``, two roundings in ``dtype`` — then float32.
    ``dtype=np.float32`` is env.py (cast at :75 precedes the scaling at :78-79; the device does the
    same arithmetic itself when given ``in_scale``/``in_offset``); ``dtype=np.float64`` is
    env_smd.py:76-77, 97 (scaled in float64, the frame cast to float32 afterwards): that variant
    is host arithmetic here as it is in the reference — once per series at load time."""
    x = np.asarray(x, dtype=dtype)
    scale, off = minmax_params(x, dtype)
    return (x * scale + off).astype(np.float32)


# ------------------------------------------------------------------------------------------ weights


def vae_from_dense_chain(chain, logvar_clip: float = 10.0) -> Dict:
    """[(kernel, bias), ...] in creation order hidden..., mu, logvar, decoder hidden, output
    (reference myasp-A1.py:89-99) -> the dict WindowScorer.set_vae takes."""
    n = len(chain)
    if n < 4:
        raise ValueError("a VAE Dense chain has at least mu, logvar, decoder hidden, output")
    mu = n - 4
    return {"kernels": [k for k, _ in chain], "biases": [b for _, b in chain],
            "acts": [1] * mu + [0, 0, 1, 2], "mu_layer": mu, "logvar_layer": mu + 1,
            "logvar_clip": logvar_clip}


_default_scorer: Optional[WindowScorer] = None


def default_scorer(device: int = 0) -> WindowScorer:
    global _default_scorer
    if _default_scorer is None:
        _default_scorer = WindowScorer(device)
    return _default_scorer


class KerasVAE:
    """Stands in for the Keras model returned by ``load_model('vae_model.h5', ...)``
    (reference myasp-A1.py:567): ``predict(x)`` -> reconstruction, float32 (B, D).

    The reference's ``predict`` samples z = mu + exp(logvar/2)*eps with fresh noise on every call
    (myasp-A1.py:78-84); this one takes the mean path z = mu unless ``eps`` is given.
    """

    def __init__(self, vae: Dict, scorer: Optional[WindowScorer] = None):
        self.config = vae
        self.scorer = scorer or default_scorer()
        self.scorer.set_vae(vae)
        self.input_dim = vae["kernels"][0].shape[0]

    def _bind(self):
        # several models may share one scorer: re-send the weights when another model was bound
        if getattr(self.scorer, "_bound_vae", None) is not self:
            self.scorer.set_vae(self.config)
            self.scorer._bound_vae = self

    def predict(self, x, batch_size=None, verbose=0, eps=None):
        self._bind()
        x = np.asarray(x, dtype=np.float32)
        if x.ndim == 1:
            x = x[None, :]
        return self.scorer.vae_predict(x, eps=eps)

    def reconstruction_error(self, x) -> np.ndarray:
        self._bind()
        x = np.asarray(x, dtype=np.float32)
        if x.ndim == 1:
            x = x[None, :]
        return self.scorer.vae_predict(x, return_err=True)[1]


def load_model(path: str, custom_objects=None, compile: bool = False,
               scorer: Optional[WindowScorer] = None) -> KerasVAE:
    """Signature of ``tensorflow.keras.models.load_model`` as the reference calls it
    (myasp-A1.py:567); ``custom_objects``/``compile`` are accepted and ignored."""
    return KerasVAE(load_keras_vae(path), scorer)


class QEstimator:
    """Stands in for ``Q_Estimator_Nonlinear`` / ``QNet`` at inference time
    (reference myasp-A1.py:165-199, myasp-wadi.py:208-229)."""

    def __init__(self, lstm: Dict, scope: str = "qlearn", scorer: Optional[WindowScorer] = None):
        self.scope = scope
        self.weights = lstm
        self.scorer = scorer or default_scorer()
        self.scorer.set_lstm(lstm)
        self.scorer._bound_lstm = self

    @classmethod
    def from_checkpoint(cls, prefix: str, scope: str = "qlearn", scorer=None) -> "QEstimator":
        return cls(load_q_estimator(prefix, scope), scope, scorer)

    def _bind(self):
        if getattr(self.scorer, "_bound_lstm", None) is not self:
            self.scorer.set_lstm(self.weights)
            self.scorer._bound_lstm = self

    def predict(self, state, sess=None):
        """``sess.run(self.action_values, {self.state: state})``: state (B, W, I) -> (B, A)."""
        self._bind()
        s = np.asarray(state, dtype=np.float32)
        if s.ndim == 2:
            s = s[None]
        return self.scorer.q_predict(s)


def vae_from_checkpoint(prefix: str, scorer=None) -> Optional[KerasVAE]:
    """The VAE Dense layers that ride along in a Saver bundle."""
    chain = load_vae_dense_chain(prefix)
    return None if chain is None else KerasVAE(vae_from_dense_chain(chain), scorer)


# ------------------------------------------------------------------------------------------ a-2


def make_state_fn(n_steps: int = 25):
    """``RNNBinaryStateFuc`` (reference myasp-A1.py:120-134): the env's pluggable state function.
    Pure array assembly, kept on the host exactly as in the reference (env.py:92, 143, 162-164
    call it one cursor at a time); the batched equivalent is the flag modes of the GPU path."""

    def RNNBinaryStateFuc(timeseries, timeseries_curser, previous_state=[], action=None):
        v = timeseries["value"]
        if timeseries_curser == n_steps:
            st = np.zeros((n_steps, 2), dtype="float32")
            st[:, 0] = np.asarray(v[1:n_steps + 1], dtype="float32")
            st[-1, 1] = 1
            return st
        if timeseries_curser > n_steps:
            prev = np.asarray(previous_state, dtype="float32")
            out = np.empty((2, n_steps, 2), dtype="float32")
            out[:, :n_steps - 1] = prev[1:n_steps]
            out[:, n_steps - 1, 0] = v[timeseries_curser]
            out[0, n_steps - 1, 1] = 0
            out[1, n_steps - 1, 1] = 1
            return out
        return None

    return RNNBinaryStateFuc


# ------------------------------------------------------------------------------------------ a-5..a-7


class RewardFunction:
    """``RNNBinaryRewardFuc`` (reference myasp-A1.py:137-151; SMD variant myasp-smd.py:136-153).

    The reference calls ``vae.predict`` on one window per ``env.step``.  Here the reconstruction
    error of *every* window of a series is computed by one fused GPU call the first time the
    series is seen and then indexed by cursor — same values, one launch per series instead of one
    Keras dispatch per step.
    """

    def __init__(self, vae: KerasVAE, n_steps: int = 25, tn=1.0, tp=5.0, fp=-1.0, fn=-5.0,
                 include_vae_penalty: bool = True, label_column: str = "label"):
        self.vae, self.W = vae, n_steps
        self.tn, self.tp, self.fp, self.fn = tn, tp, fp, fn
        self.include = include_vae_penalty
        self.label_column = label_column
        self._cache: Dict[tuple, np.ndarray] = {}
        self.max_cached_series = 256
        self._last = None          # (values the errors were computed from, errors, vae identity) of the last series

    def clear(self):
        """Forget every cached series (call after re-training the VAE in place)."""
        self._cache.clear()
        self._last = None

    def error_at(self, timeseries, c: int) -> float:
        """The reconstruction error of the window the reward at cursor ``c`` looks at, v[c-W : c].  The reference
        calls the reward function at every env.step; hashing the whole series each time (errors_for) would make
        the cache lookup O(N) per step.  The error depends on those W values only, so the per-step path compares
        just them with the series the last lookup scored — O(W), and exact: a different series, or one edited in
        place inside the window, falls through to the content-keyed lookup."""
        W = self.W
        col = np.asarray(timeseries["value"])                       # a view for ndarray / pandas columns
        last = self._last
        if last is not None and last[2] == (id(self.vae), id(self.vae.config)) and len(col) == len(last[0]) \
                and np.array_equal(np.asarray(col[c - W:c], dtype=np.float32), last[0][c - W:c]):
            return float(last[1][c - W])
        return float(self.errors_for(timeseries)[c - W])

    def errors_for(self, timeseries) -> np.ndarray:
        # keyed on the CONTENT of the value column (length + a hash of its bytes) and on the VAE the errors
        # were computed with: an id() can be reused after garbage collection, a frame can be edited in place
        v = np.ascontiguousarray(np.asarray(timeseries["value"], dtype=np.float32))
        key = (len(v), hash(v.tobytes()), id(self.vae), id(self.vae.config))
        err = self._cache.get(key)
        if err is None:
            self.vae._bind()
            err = self.vae.scorer.score(v, self.W, vae_lag=1, want=("err",))["err"]
            if len(self._cache) >= self.max_cached_series:
                self._cache.pop(next(iter(self._cache)))            # oldest entry
            self._cache[key] = err
        self._last = (v.copy(), err, (id(self.vae), id(self.vae.config)))      # a copy: v may alias the caller's column
        return err

    def __call__(self, timeseries, timeseries_curser, action=0, vae=None, dynamic_coef=1.0):
        c = timeseries_curser
        if c < self.W:
            return [0, 0]
        pen = dynamic_coef * self.error_at(timeseries, c) if self.include else 0.0
        lab = timeseries[self.label_column][c]
        if lab == 0:
            return [self.tn + pen, self.fp + pen]
        if lab == 1:
            return [self.fn + pen, self.tp + pen]
        return [0, 0]


def update_dynamic_coef_proportional(current_coef, episode_reward, target_reward=100.0, alpha=0.01, min_coef=0.1,
                                     max_coef=10.0):
    """The dynamic reward coefficient's per-episode update (reference myasp-A1.py:235-238; called with
    target_reward=0.0, alpha=0.001 at :455): scalar host arithmetic, same signature."""
    return max(min(current_coef + alpha * (target_reward - episode_reward), max_coef), min_coef)


def penalty_array(values, n_steps: int, vae: KerasVAE, center=None, scale=None, clip: float = 10.0):
    """The module-level ``penalty_array`` of the WaDi driver (reference myasp-wadi.py:130-149):
    ``values`` (N, F) raw sensors -> float array (N,), zeros for the first n_steps-1 entries, then
    the reconstruction error of the window ending at each sample.  ``center`` / ``scale`` are the
    fitted ``RobustScaler``'s ``center_`` / ``scale_`` (they cross the C-ABI in float64 whatever their
    dtype); the device applies sklearn's in-place float32 transform and the clip."""
    x = np.asarray(values, dtype=np.float32)
    if x.ndim == 1:
        x = x[:, None]
    N = x.shape[0]
    out = np.zeros(N, dtype=np.float64)
    if N < n_steps:
        return out
    vae._bind()
    # window ending at sample i (i >= W-1) = window index k = i - W of a series with one
    # leading pad row; score the padded series with lag 0.
    padded = np.concatenate([x[:1], x], axis=0)
    err = vae.scorer.score(padded, n_steps, vae_lag=0, want=("err",), vae_center=center,
                           vae_scale=scale, vae_clip=clip if center is not None else 0.0)["err"]
    out[n_steps - 1:] = err
    return out


# ------------------------------------------------------------------------------------------ a-10


class active_learning:
    """``active_learning`` of the drivers (reference myasp-A1.py:242-259): same constructor and
    ``get_samples`` contract, but the margins of all windows come from one GPU call on the env's
    current series instead of one ``estimator.predict`` per state."""

    def __init__(self, env, N, strategy, estimator: QEstimator, already_selected, n_steps: int = 25):
        self.env, self.N, self.strategy = env, N, strategy
        self.estimator, self.already_selected = estimator, already_selected
        self.W = n_steps

    def distances(self) -> np.ndarray:
        self.estimator._bind()
        v = np.asarray(self.env.timeseries["value"], dtype=np.float32)
        # env.states_list carries the get_states_list flag quirk
        return self.estimator.scorer.score(v, self.W, flag_mode="quirk", want=("margin",))["margin"]

    def get_samples(self) -> List[int]:
        self.estimator._bind()
        v = np.asarray(self.env.timeseries["value"], dtype=np.float32)
        # scoring, ranking and the already_selected mask all happen on the device
        return self.estimator.scorer.al_get_samples(v, self.W, self.N, self.already_selected, flag_mode="quirk")

    def get_samples_by_score(self, threshold) -> List[int]:
        d = self.distances()
        sel = set(int(i) for i in self.already_selected)
        return [int(i) for i in np.argsort(d) if int(i) not in sel and d[i] < threshold]
