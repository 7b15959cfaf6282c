"""bench.py as the driver runs it: a bare `python bench.py --gpus N` must start its own ranks.  On the
one-GPU box the N > 1 path is rehearsed (DRLVAE_BENCH_REHEARSE=1: every rank on device 0, gloo instead of
RCCL, which refuses two ranks on one device) — the code path of step()/fence()/ScorePack, never a measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=280)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_default_line_has_the_contract_fields():
    d = _run(["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alt"])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["unit"] == "windows/s"
    assert d["config"]["workload"].startswith("C2") and d["config"]["windows_per_gpu_per_step"] == 93191
    r = d["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert d["value"] > 1e7
    assert d["config"]["lstm_kernels"] == ["lstm_h128i2_kernel<2>"] and d["config"]["rccl_ranks"] == 1
    # traffic follows from the committed PMC summary of exactly the kernels that ran
    assert r["traffic"] is None or r["traffic"] > 0
    lib = d["library"]
    assert lib["stale"] is False and lib["src_hash_compiled_in"] == lib["src_hash_of_tree"] and "abi 2" in lib["version"]
    # the driver's one line carries every other BASELINE config and the host-pointer drop-in path
    oc = {c["workload"][:2]: c for c in d["other_configs"]}
    assert sorted(oc) == ["C1", "C3", "C4", "C5"]
    for c in oc.values():
        for key in ("value", "ms_per_step", "lstm_kernel", "lstm_kernels", "frac", "frac_alone", "vae_kernel_ms_alone", "steps"):
            assert key in c, key
        assert 0 < c["frac"] < 1 and 0 < c["frac_alone"] < 1 and c["value"] > 1e6
    assert oc["C5"]["timed_region_s"] >= 0.8 and oc["C5"]["windows_per_step"] == 4_000_000     # a sustained leg
    assert oc["C3"]["lstm_kernel"] == oc["C4"]["lstm_kernel"] == "xproj+rec" and oc["C4"]["vae_kernel"] == "wide16s"
    assert 0 < oc["C4"]["vae_frac_alone"] < 1
    hp = {h["workload"][:2]: h for h in d["host_path"]}
    assert sorted(hp) == ["C2", "C4"] and all(h["value"] > 1e6 and h["ms_per_call"] > 0 for h in hp.values())


def test_two_ranks_from_a_bare_invocation_weak_c5_and_sharded_c4():
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "2e5", "--no-alt"], {"DRLVAE_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["workload"].startswith("C5")
    assert d["config"]["windows_per_rank"] == [200000, 200000] and d["config"]["rccl_ranks"] == 2
    assert "REHEARSAL" in d["config"]["rccl_ranks_is"] and len(d["config"]["devices"]) == 2
    assert "single_gpu_same_workload" in d
    d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "c4", "--shard", "--windows", "20000"],
             {"DRLVAE_BENCH_REHEARSE": "1"})
    assert d["scaling"] == "strong" and d["config"]["windows_per_rank"] == [10000, 10000]
    assert d["config"]["windows_per_step_all_gpus"] == 20000
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["algorithmic_frac"] > d["roofline"]["frac"]
