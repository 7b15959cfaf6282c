"""ctypes binding of libdrlvae.so (include/drlvae.h).  No fallback: when the library or a GPU is
missing every entry point raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRLVAE_LIB points experiments (ablation builds) at another build of the same library
LIB_PATH = os.environ.get("DRLVAE_LIB") or os.path.join(_HERE, "libdrlvae.so")

FLAG_ZEROS, FLAG_QUIRK, FLAG_EXPLICIT, FLAG_BOTH = 0, 1, 2, 3
ACTION_ARGMAX, ACTION_WADI = 0, 1
COMM_ID_BYTES = 128

# every symbol include/drlvae.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "drl_create", "drl_destroy", "drl_last_error", "drl_version", "drl_set_stream", "drl_synchronize",
    "drl_set_vae", "drl_set_lstm", "drl_set_lstm_math", "drl_score_batch", "drl_score_batch_dev", "drl_score_batch_dev_noalloc", "drl_score_series",
    "drl_vae_predict", "drl_q_predict", "drl_rollout_batch", "drl_set_rollout_mode", "drl_al_get_samples", "drl_profile_enable", "drl_profile_reset", "drl_profile_read",
    "drl_last_lstm_variant", "drl_last_lstm_kernels", "drl_last_vae_variant", "drl_last_lstm_tile", "drl_last_call_overlapped",
    "drl_comm_unique_id", "drl_comm_init", "drl_allgather_scores", "drl_comm_destroy", "drl_comm_size",
    "drl_set_knob", "drl_debug_alloc_count",
]
ABI_REVISION = 2          # DRL_ABI_REVISION of include/drlvae.h

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class ScoreOpts(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("vae_lag", C.c_int32),
        ("flag_mode", C.c_int32),
        ("action_rule", C.c_int32),
        ("flags", C.c_void_p),
        ("eps", C.c_void_p),
        ("in_scale", C.c_void_p),
        ("in_offset", C.c_void_p),
        ("vae_center", C.c_void_p),
        ("vae_scale", C.c_void_p),
        ("vae_clip", C.c_float),
        ("abi_revision", C.c_int32),
    ]


_lib = None


def _one_hip_runtime():
    """A process can hold one HIP/HSA runtime.  PyTorch-ROCm wheels ship their own copy
    (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7) and look it up by a file name the system
    copy does not answer to, so a process that loads libdrlvae.so first and torch later would end up
    with two runtimes and torch would see no GPU.  When torch is installed but not yet imported, map
    its copy first: libdrlvae.so's NEEDED libamdhip64.so.7 then resolves to it by SONAME, whichever
    order the caller imports things in.  Without torch the system runtime (RUNPATH /opt/rocm) is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """Loads libdrlvae.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the scoring path.")
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.drl_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.drl_destroy.argtypes = [vp]
    lib.drl_destroy.restype = None
    lib.drl_last_error.argtypes = [vp]
    lib.drl_last_error.restype = C.c_char_p
    lib.drl_version.restype = C.c_char_p
    lib.drl_set_stream.argtypes = [vp, vp]
    lib.drl_synchronize.argtypes = [vp]
    lib.drl_set_lstm_math.argtypes = [vp, C.c_int]
    lib.drl_set_rollout_mode.argtypes = [vp, C.c_int]
    lib.drl_set_vae.argtypes = [vp, C.c_int, _i32p, _i32p, _i32p, C.c_int, C.c_int,
                                C.POINTER(vp), C.POINTER(vp), C.c_float]
    lib.drl_set_lstm.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp]
    score_args = [vp, vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(ScoreOpts), vp, vp, vp, vp]
    lib.drl_score_batch.argtypes = score_args
    lib.drl_score_batch_dev.argtypes = score_args
    lib.drl_score_batch_dev_noalloc.argtypes = score_args
    lib.drl_score_series.argtypes = [vp, vp, C.c_int64, C.c_int, C.c_int, C.POINTER(ScoreOpts), vp, vp, vp, vp]
    lib.drl_vae_predict.argtypes = [vp, vp, C.c_int64, vp, vp, vp]
    lib.drl_q_predict.argtypes = [vp, vp, C.c_int64, C.c_int, vp]
    lib.drl_rollout_batch.argtypes = [vp, vp, _i64p, C.c_int, C.c_int, vp, vp, vp, vp, _i32p]
    lib.drl_al_get_samples.argtypes = [vp, vp, _i64p, C.c_int, C.c_int, C.c_int, C.POINTER(ScoreOpts), vp, C.c_int,
                                       C.c_int, vp, _i32p, vp]
    lib.drl_profile_enable.argtypes = [vp, C.c_int]
    lib.drl_profile_reset.argtypes = [vp]
    lib.drl_profile_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), _i64p]
    lib.drl_last_lstm_variant.argtypes = [vp]
    lib.drl_last_lstm_variant.restype = C.c_char_p
    lib.drl_last_lstm_kernels.argtypes = [vp]
    lib.drl_last_lstm_kernels.restype = C.c_char_p
    lib.drl_last_vae_variant.argtypes = [vp]
    lib.drl_last_vae_variant.restype = C.c_char_p
    lib.drl_last_lstm_tile.argtypes = [vp]
    lib.drl_last_lstm_tile.restype = C.c_int
    lib.drl_last_call_overlapped.argtypes = [vp]
    lib.drl_comm_unique_id.argtypes = [vp, vp]
    lib.drl_comm_init.argtypes = [vp, C.c_int, C.c_int, vp]
    lib.drl_allgather_scores.argtypes = [vp, vp, vp, C.c_int64]
    lib.drl_comm_destroy.argtypes = [vp]
    lib.drl_comm_size.argtypes = [vp, _i32p]
    lib.drl_set_knob.argtypes = [vp, C.c_char_p, C.c_longlong]
    lib.drl_debug_alloc_count.argtypes = []
    lib.drl_debug_alloc_count.restype = C.c_longlong
    _lib = lib
    return lib
