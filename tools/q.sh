#!/bin/bash
# Quick per-workload bench on the GPU box: tools/q.sh <tag> "c4 8 2" "c3 40 5" ...  (workload steps warmup);
# DRLVAE_LIB selects another build of the library.  Prints one summary line per workload.
tag=$1; shift
mkdir -p gpurun_out/q
for w in "$@"; do
  set -- $w
  f=gpurun_out/q/${tag}_$1
  timeout -k 10 150 python -u bench.py --workload $1 --steps $2 --warmup $3 --no-cpu-baseline --no-alt > $f.json 2> $f.err || { echo "$tag $1 FAILED"; tail -5 $f.err; exit 1; }
  python - "$f.json" "$tag" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; a = r.get("kernels_alone") or {}
al = f"alone lstm {a['lstm_kernels_avg_ms']:.3f} vae {a['vae_kernel_avg_ms']:.3f}" if a else ""
print(f"{sys.argv[2]} {d['config']['workload'][:2]}: {d['value']/1e6:.2f} M/s  step {d['ms_per_step']:.3f} ms  lstm {r['avg_launch_ms']:.3f}  vae {r['vae_kernel_avg_ms']:.3f}  frac {r['frac']:.3f}  {al}")
PY
done
