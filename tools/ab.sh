#!/bin/bash
# Runs bench.py once per library variant (build/libdrlvae_<name>.so) and prints LSTM kernel time.
# usage: tools/ab.sh <workload> <windows> name1 name2 ...
wl=$1; win=$2; shift 2
for name in "$@"; do
  DRLVAE_LIB=$PWD/build/libdrlvae_$name.so timeout -k 10 200 python -u bench.py --workload $wl --windows $win --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/ab_$name.err; exit 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$name.json"))
r=d["roofline"]
print(f"$name: {d['value']/1e6:.2f} Mwin/s  lstm {r['avg_launch_ms']:.3f} ms  frac {r['frac']:.3f}  vae {r['vae_kernel_avg_ms']:.3f} ms")
PY
done
