"""Runs the native C-ABI caller (tests/native/abi_driver.cpp) whose host code — the library's
included — is built under AddressSanitizer + UBSan by tests/native/Makefile
(__graft_entry__.build() makes it; the GPU box only runs the prebuilt binary)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "native", "_build", "abi_driver")


def test_native_abi_driver_under_host_sanitizers():
    if not os.path.exists(BIN):
        pytest.skip(f"{BIN} is missing: `make -C tests/native` (hipcc; __graft_entry__.build() runs it) builds it")
    env = dict(os.environ)
    # leak checking would report the HIP runtime's own process-lifetime allocations
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([BIN], env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, f"rc={r.returncode}\nstdout:\n{r.stdout[-2000:]}\nstderr:\n{r.stderr[-4000:]}"
    assert "abi_driver OK" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-4000:]
