#!/bin/bash
# Builds timing-only ablation variants of libdrlvae.so into build/ (see H128_ABL in kernels.hip.h).
# usage: tools/build_variants.sh "name1:-DFLAG=1 -DOTHER=2" "name2:..."
set -e
cd "$(dirname "$0")/../dynamic-reward-rl-vae_amd/csrc"
mkdir -p ../../build
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1 $flags \
      -o ../../build/libdrlvae_$name.so drlvae.hip &
done
wait
ls -la ../../build/
