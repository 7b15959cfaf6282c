#!/usr/bin/env python3
"""Copies the round's profile results (gpurun_out/round) into profiles/rNN and writes the PMC summary
that bench.py reads for roofline.traffic.  usage: tools/summarise_profiles.py r1"""
import collections, csv, glob, json, os, shutil, sys
rnd = sys.argv[1]
src, dst = "gpurun_out/round", f"profiles/{rnd}"
os.makedirs(dst, exist_ok=True)
for f in glob.glob(f"{src}/bench_*.json"):
    shutil.copy(f, dst)
for w in ("c2", "c3", "c4"):
    fs = sorted(glob.glob(f"{src}/trace_{w}/*/*kernel_stats.csv"), key=os.path.getmtime)
    if fs:                                  # gpurun merges results: older runs may still lie beside the newest
        shutil.copy(fs[-1], f"{dst}/{w}_kernel_stats.csv")
for w in ("c2", "c3", "c4"):
    out = {"workload": w,
           "command": f"rocprofv3 --pmc <group> --output-format csv -- python bench.py --workload {w} --steps 5 --warmup 1 --no-cpu-baseline --no-alt --no-others (one run per counter group, so one launch population per kernel; tools/profile_round.sh)",
           "note": "per-launch averages; FETCH_SIZE/WRITE_SIZE in KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide coalesced read stream, so hbm_read_bytes = 2*FETCH_SIZE*1024 (an upper bound for these kernels' dword loads); WRITE_SIZE is exact.",
           "kernels": {}}
    for d in ("sq", "fetch", "write", "inst"):
        for f in sorted(glob.glob(f"{src}/pmc_{w}_{d}/*/*counter_collection.csv"), key=os.path.getmtime)[-1:]:
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in agg.items():
                if k.startswith("__amd"):
                    continue
                for c, vals in v.items():
                    out["kernels"].setdefault(k, {})[c] = sum(vals) / len(vals)
    if not out["kernels"]:
        continue
    for k, v in out["kernels"].items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_read_bytes_corrected"] = 2 * v["FETCH_SIZE"] * 1024
            v["hbm_write_bytes"] = v["WRITE_SIZE"] * 1024
            v["hbm_bytes_per_launch"] = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            v["mfma_util_pct"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8 * 1024) * 100
    json.dump(out, open(f"{dst}/{w}_pmc_summary.json", "w"), indent=1)
    for k, v in out["kernels"].items():
        print(w, k, {c: round(x, 1) for c, x in v.items() if c in ("mfma_util_pct", "hbm_bytes_per_launch", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE")})
