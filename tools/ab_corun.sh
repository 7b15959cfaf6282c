#!/bin/bash
# Can one lstm_rec workgroup (256 registers) and one vae_wide16 workgroup share a CU and fill each other's
# MFMA bubbles?  build/libdrlvae_rw2.so = -DREC_WAVES_PER_SIMD=2; DRLVAE_REC_LDS_PAD sizes the rec workgroup's
# LDS so that two of them do not fit a CU but one of them and one VAE workgroup do.
run() { # label lib pad
  export DRLVAE_LIB=$2
  if [ "$3" = 0 ]; then unset DRLVAE_REC_LDS_PAD; else export DRLVAE_REC_LDS_PAD=$3; fi
  timeout -k 10 200 python -u bench.py --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/corun.json 2> gpurun_out/corun.err || { echo "$1 FAILED"; tail -3 gpurun_out/corun.err; exit 1; }
  python - "$1" <<'PY'
import json,sys
d=json.load(open("gpurun_out/corun.json"))
print(sys.argv[1], f"{d['value']/1e6:.2f} Mwin/s  step {d['ms_per_step']:.3f} ms", d["config"]["vae_and_lstm_kernels_overlap"])
PY
}
L=$PWD/dynamic-reward-rl-vae_amd/drlvae/libdrlvae.so
R=$PWD/build/libdrlvae_rw2.so
run "default" $L 0 && run "default, rec 1 WG/CU by LDS" $L 18000 && run "rec 256 regs" $R 0 && run "rec 256 regs, 1 rec WG/CU by LDS (co-residency with a VAE WG possible)" $R 18000 && DRLVAE_OVERLAP=0 run "rec 256 regs, 1 WG/CU, no overlap" $R 18000
