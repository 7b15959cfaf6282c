#!/bin/bash
# Timeline (kernels + memory copies) of drl_score_batch on C4: rocprofv3 traces of tools/host_path_rate.py, last call summarised
export TMPDIR=/tmp
out=gpurun_out/hp_trace; rm -rf $out; mkdir -p $out
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python tools/host_path_rate.py c4 > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python - "$out" <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:28]))
for f in glob.glob(sys.argv[1] + "/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", ""))[:20]))
ev.sort()
# the last call: events after the last gap of > 1 ms ... take the final 60 events
last = ev[-70:]
t0 = last[0][0]
for s, e, n in last:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {n}")
PY
