#!/bin/bash
# Runs bench.py for one workload under several settings of one environment variable.
# usage: tools/ab_env.sh <workload> <steps> VAR v1 v2 ...
wl=$1; steps=$2; var=$3; shift 3
for v in "$@"; do
  env $var=$v timeout -k 10 200 python -u bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline > gpurun_out/abe_$v.json 2> gpurun_out/abe_$v.err || { echo "$var=$v FAILED"; tail -3 gpurun_out/abe_$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/abe_$v.json"))
r=d["roofline"]
print(f"$wl $var=$v: {d['value']/1e6:.2f} Mwin/s  step {d['ms_per_step']:.3f} ms  lstm {r['avg_launch_ms']:.3f} ms  frac {r['frac']:.3f}  vae {r['vae_kernel_avg_ms']:.3f} ms")
PY
done
