#!/bin/bash
# Quick loop on the GPU box: parity suite, then the bench line (key figures printed).
#   gpurun --timeout 600 -- 'bash tools/gpu_quick.sh TAG [bench args]'
set -u
TAG=${1:-quick}; shift || true
mkdir -p gpurun_out/r02
python -m pytest tests -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -4
python bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" 2>/dev/null | tail -1 > gpurun_out/r02/${TAG}_bench.json
python - <<PY
import json
d = json.loads(open("gpurun_out/r02/${TAG}_bench.json").read())
r = d["roofline"]
print("Gpx/s %.1f  ms/step %.4f  K1 %.4f ms  all kernels %.4f ms  frac %.4f  bit_exact %s" % (
    d["value"] / 1e3, d["ms_per_step"], r["kernel_ms"], r["all_kernels_ms"], r["frac"], d.get("bit_exact")))
PY
