#!/usr/bin/env python3
"""Prints one line per bench.py JSON file given on the command line."""
import json
import sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{f}: {d['value'] / 1e6:.2f} Mwin/s  ms/step {d['ms_per_step']:.3f}  lstm {r['avg_launch_ms']:.3f} ms  "
          f"vae {r['vae_kernel_avg_ms']:.3f} ms  frac {r['frac']:.3f}  {d['config']['lstm_kernel']}")
