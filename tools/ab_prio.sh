#!/bin/bash
# overlap x aux-stream priority on C3, C4, C2
run() { # wl steps overlap prio
  if [ "$4" = none ]; then unset DRLVAE_AUX_PRIO; else export DRLVAE_AUX_PRIO=$4; fi
  DRLVAE_OVERLAP=$3 timeout -k 10 200 python -u bench.py --workload $1 --steps $2 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/prio.json 2> gpurun_out/prio.err || { echo FAILED; tail -3 gpurun_out/prio.err; exit 1; }
  python - "$1 overlap=$3 prio=$4" <<'PY'
import json,sys
d=json.load(open("gpurun_out/prio.json"))
print(sys.argv[1], f"{d['value']/1e6:.2f} Mwin/s  step {d['ms_per_step']:.3f} ms")
PY
}
for wl in "c3 60" "c4 10" "c2 20"; do
  set -- $wl
  run $1 $2 0 none && run $1 $2 2 none && run $1 $2 2 low && run $1 $2 2 high || exit 1
done
