"""CPU-only: libdrlvae.so loads, exports every symbol include/drlvae.h declares, and the struct
layout the Python side uses matches the header.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _header():
    return open(os.path.join(ROOT, "include", "drlvae.h")).read()


def test_library_exports_every_declared_symbol():
    from drlvae import _abi
    lib = _abi.load()
    declared = set(re.findall(r"\b(drl_[a-z0-9_]+)\s*\(", _header()))
    assert declared == set(_abi.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.drl_version()


def test_library_is_built_from_the_sources_beside_it():
    """drl_version carries the sha256 of the sources the library was compiled from (csrc/Makefile): a prebuilt
    libdrlvae.so that no longer matches the tree fails here, before it can pass a GPU test unnoticed."""
    import sys
    from drlvae import _abi
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from src_hash import src_hash
    v = _abi.load().drl_version().decode()
    m = re.search(r"src ([0-9a-f]{16})\)", v)
    assert m, v
    assert m.group(1) == src_hash(), f"{v}: stale library, rebuild (python -c 'import __graft_entry__ as g; g.build()')"
    assert f"abi {_abi.ABI_REVISION};" in v


def test_opts_struct_layout_matches_header():
    from drlvae import _abi
    body = re.search(r"typedef struct drl_score_opts \{(.*?)\} drl_score_opts;", _header(), re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:const\s+)?(int32_t|float|double)\s*(\*?)\s*(\w+)\s*;", body)
    assert [f[2] for f in fields] == [n for n, _ in _abi.ScoreOpts._fields_]
    for (typ, star, name), (_, ctype) in zip(fields, _abi.ScoreOpts._fields_):
        want = C.c_void_p if star else {"int32_t": C.c_int32, "float": C.c_float}[typ]
        assert ctype is want, name
    assert C.sizeof(_abi.ScoreOpts) == 72


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from drlvae.scorer import WindowScorer
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device|failed"):
        WindowScorer(0)


def test_header_is_plain_c(tmp_path):
    """include/drlvae.h is the boundary a C host binds: it must compile as C, not only as C++."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text('#include "drlvae.h"\nint main(void) { drl_score_opts o; o.struct_size = (int)sizeof o; '
                   'return o.struct_size == 72 ? 0 : 1; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    exe = tmp_path / "use"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", f"-I{inc}", str(src), "-o", str(exe),
                    "-Wl,--unresolved-symbols=ignore-all"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0          # struct layout = what _abi.ScoreOpts declares


def _integration_snippet():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. C level"):text.index("## 3. Entry point")]
    return re.search(r"```python\n(.*?)```", sec, re.S).group(1)


def test_integration_md_ctypes_example_declares_what_it_calls():
    """INTEGRATION.md section 2 is executable: its declarations run here (no GPU needed) and every entry point
    it calls has argtypes (a bare Python int would be passed as a 32-bit int and truncate a pointer)."""
    from drlvae import _abi
    src = _integration_snippet()
    ns = {"LIB_PATH": _abi.LIB_PATH}
    exec(compile(src, "INTEGRATION.md#2", "exec"), ns)
    lib = ns["lib"]
    called = set(re.findall(r"lib\.(drl_[a-z_]+)\(", src))
    assert {"drl_create", "drl_set_vae", "drl_set_lstm", "drl_score_series", "drl_destroy", "drl_last_error"} <= called
    for name in called:
        assert getattr(lib, name).argtypes is not None, name
    assert lib.drl_last_error.restype is C.c_char_p
    assert C.sizeof(ns["ScoreOpts"]) == C.sizeof(_abi.ScoreOpts)
    for (n1, t1), (n2, t2) in zip(ns["ScoreOpts"]._fields_, _abi.ScoreOpts._fields_):
        assert n1 == n2 and t1 is t2


@pytest.mark.gpu
def test_integration_md_ctypes_example_runs(a1, ckpt_a1):
    import numpy as np
    import torch  # noqa: F401  a process that also uses torch imports it BEFORE the library
    from drlvae import _abi
    from drlvae.scorer import WindowScorer
    ns = {"LIB_PATH": _abi.LIB_PATH}
    exec(compile(_integration_snippet(), "INTEGRATION.md#2", "exec"), ns)
    lstm, vae = ckpt_a1
    f = lambda a: np.ascontiguousarray(a, np.float32)
    v = f(a1["series"][0])
    err, q, act, margin = ns["run"](v, [f(k) for k in vae["kernels"]], [f(b) for b in vae["biases"]], vae["acts"],
                                    vae["mu_layer"], vae["logvar_layer"], f(lstm["kernel"]), f(lstm["bias"]),
                                    f(lstm["w_out"]), f(lstm["b_out"]))
    sc = WindowScorer(0)
    try:
        sc.set_vae(vae)
        sc.set_lstm(lstm)
        want = sc.score(v, 25, flag_mode="quirk")
    finally:
        sc.close()
    for got, key in ((err, "err"), (q, "q"), (act, "action"), (margin, "margin")):
        assert np.array_equal(got.view(np.uint32), want[key].view(np.uint32)), key
