#!/bin/bash
# Two SQ counter passes over one bench.py workload (run on the GPU box); prints per-kernel averages.
# usage: tools/pmc_one.sh <workload> [extra bench args]
set -e
w=$1; shift
out=gpurun_out/pmc1; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
args="--workload $w --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-others $*"
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $out/sq -- python bench.py $args > $out/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $out/inst -- python bench.py $args > $out/inst.log 2>&1
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc1/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if k.startswith("__amd") or "finalize" in k:
        continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    line = f"{k}: "
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line += f"cycles {cyc:.0f} ({cyc / 2.4e6:.3f} ms @2.4GHz)  mfma_busy {m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024) * 100:.1f}%  "
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if wc:
        line += "  ".join(f"{c[3:]} {m[c] / wc * 100:.1f}%" for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if c in m)
    print(line)
    print("    " + "  ".join(f"{c[3:]} {m[c]:.3g}" for c in sorted(m) if c.startswith("SQ_INST") or c.startswith("SQ_LDS") or c.startswith("SQ_ACTIVE_INST_")))
PY
