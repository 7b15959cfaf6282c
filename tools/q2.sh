#!/bin/bash
# tools/q2.sh <knob=value> <workload> <windows> : VAE kernel time alone for one knob setting (env DRLVAE_<KNOB>)
kv=$1; wl=$2; win=$3
var=DRLVAE_$(echo ${kv%%=*} | tr a-z A-Z); val=${kv#*=}
mkdir -p gpurun_out/q
env $var=$val DRLVAE_OVERLAP=0 timeout -k 10 150 python -u bench.py --workload $wl --windows $win --steps 10 --warmup 2 --no-cpu-baseline --no-alt --no-others > gpurun_out/q/k.json 2> gpurun_out/q/k.err || { echo FAILED; tail -5 gpurun_out/q/k.err; exit 1; }
python - "$kv" <<'PY'
import json, sys
d = json.load(open("gpurun_out/q/k.json")); r = d["roofline"]
print(f"{sys.argv[1]} {d['config']['workload'][:2]} K={d['config']['windows_per_gpu_per_step']}: vae {r['vae_kernel_avg_ms']:.4f} ms ({d['config']['vae_kernel']})  lstm {r['avg_launch_ms']:.3f}")
PY
