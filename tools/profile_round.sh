#!/bin/bash
# Runs on the GPU box: bench of every workload, rocprofv3 kernel stats for C2/C4 and PMC passes for C2/C4.
# Results land in gpurun_out/round/ ; copy what matters into profiles/rNN/.
# usage: tools/profile_round.sh [bench]   with `bench`: only the bench lines (run it after the PMC summaries of the
# round are in profiles/rNN, so that every line's roofline.traffic follows from the committed summary)
set -e
mode=${1:-all}
out=gpurun_out/round; [ "$mode" = all ] && rm -rf $out; mkdir -p $out
export TMPDIR=/tmp
# the driver's flag-less run: C2 as `value`, every other config, the host-pointer path and the CPU baseline in ONE line
timeout -k 10 300 python -u bench.py > $out/bench_default.json 2> $out/bench_default.err
timeout -k 10 300 python -u bench.py --workload c2 --no-cpu-baseline > $out/bench_c2.json 2> $out/bench_c2.err
# short steps need more of them for a stable figure (C1: 0.14 ms per step, C3: 0.75 ms)
for spec in "c1 300 20" "c3 40 5" "c4 8 2" "c5 5 2"; do
  set -- $spec
  timeout -k 10 300 python -u bench.py --workload $1 --steps $2 --warmup $3 --no-cpu-baseline > $out/bench_$1.json 2> $out/bench_$1.err
done
# the n_hidden_dim = 128 variants of the multivariate configurations (C3 "H128")
timeout -k 10 300 python -u bench.py --workload c3 --hidden 128 --steps 40 --warmup 5 --no-cpu-baseline > $out/bench_c3_h128.json 2> $out/bench_c3_h128.err
timeout -k 10 300 python -u bench.py --workload c4 --hidden 128 --steps 8 --warmup 2 --no-cpu-baseline > $out/bench_c4_h128.json 2> $out/bench_c4_h128.err
if [ "$mode" = all ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c2 -- python bench.py --no-cpu-baseline --no-alt --no-others > $out/trace_c2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c4 -- python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline --no-alt --no-others > $out/trace_c4.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_c3 -- python bench.py --workload c3 --steps 20 --warmup 2 --no-cpu-baseline --no-alt --no-others > $out/trace_c3.log 2>&1
# PMC passes, one counter group per run (a run may hold at most 8 SQ / 4 TCC counters): C2, C3 and C4
for w in c2 c3 c4; do
  args="--workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-alt --no-others"
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $out/pmc_${w}_sq -- python bench.py $args > $out/pmc_${w}_sq.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_${w}_fetch -- python bench.py $args > $out/pmc_${w}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_${w}_write -- python bench.py $args > $out/pmc_${w}_write.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_${w}_inst -- python bench.py $args > $out/pmc_${w}_inst.log 2>&1
done
fi
for f in $out/bench_*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("bench_")[-1][:-5], f"{d['value']/1e6:.2f} Mwin/s", f"lstm {r['avg_launch_ms']:.3f} ms frac {r['frac']:.3f}", f"vae {r['vae_kernel_avg_ms']:.3f} ms", d["config"]["lstm_kernel"], d.get("cpu_baseline",{}).get("value"))
PY
done
