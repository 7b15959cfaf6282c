#!/bin/bash
# tools/kstat.sh <workload> [steps] [extra bench args]: rocprofv3 per-kernel average durations of one bench workload, overlap off
w=$1; st=${2:-10}; shift; shift
export TMPDIR=/tmp
out=gpurun_out/kstat_$w; rm -rf $out; mkdir -p $out
DRLVAE_OVERLAP=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python bench.py --workload $w --steps $st --warmup 2 --no-cpu-baseline --no-alt --no-others "$@" > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python - "$out" <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if "at::" in n or "amd_rocclr" in n: continue
    print(f"  {n:34s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
