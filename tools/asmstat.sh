#!/bin/bash
# Compiles csrc/drlvae.hip with extra flags and prints instruction statistics of one kernel.
# usage: tools/asmstat.sh <kernel-symbol-substring> [flags...]
k=$1; shift
d=$(mktemp -d /tmp/asmstat.XXXX)
cd "$(dirname "$0")/../dynamic-reward-rl-vae_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -save-temps=obj -c -o $d/x.o drlvae.hip 2>$d/err || { cat $d/err; exit 1; }
S=$d/drlvae-hip-amdgcn-amd-amdhsa-gfx950.s
sym=$(grep -o "^_Z[A-Za-z0-9_]*$k[A-Za-z0-9_]*:" $S | head -1 | tr -d :)
awk "/^$sym:/,/s_endpgm/" $S > $d/k.s
echo "$sym flags=[$*] lines=$(wc -l < $d/k.s) mfma=$(grep -c v_mfma $d/k.s) acc_rd=$(grep -c v_accvgpr_read $d/k.s) acc_wr=$(grep -c v_accvgpr_write $d/k.s) scratch=$(grep -c scratch_ $d/k.s) exp=$(grep -c v_exp_f32 $d/k.s)"
grep -A25 "^\s*\.amdhsa_kernel $sym" $S | grep -E "next_free_vgpr|accum_offset" | tr -s ' ' | tr '\n' ' '; echo
echo $d/k.s
