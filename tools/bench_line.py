#!/usr/bin/env python3
"""Print the key figures of bench.py JSON lines read from stdin (one per line)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line); r = d["roofline"]
    print("%s  Gpx/s %.1f  ms/step %.4f  K1 %.4f ms  all kernels %.4f ms  frac %.4f  bit_exact %s" % (
        " ".join(sys.argv[1:]), d["value"] / 1e3, d["ms_per_step"], r["kernel_ms"], r["all_kernels_ms"], r["frac"], d.get("bit_exact")))
