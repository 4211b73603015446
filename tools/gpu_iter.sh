#!/bin/bash
# One iteration of the K1 work on the GPU box: parity suite, bench line (key figures), VALU per wave of K1 per
# ablation level for the struct (bench) and noise workloads.
#   gpurun --timeout 900 -- 'bash tools/gpu_iter.sh TAG [quick]'
set -u
TAG=${1:-iter}; MODE=${2:-full}
OUT=gpurun_out/r05/$TAG; mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee $OUT/gputests.txt
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench.json
python - <<PY | tee $OUT/bench_key.txt
import json
d = json.loads(open("$OUT/bench.json").read())
r = d["roofline"]
print("Gpx/s %.1f  ms/step %.4f  K1 %.4f ms (min %.4f)  all kernels %.4f ms  frac %.4f  bit_exact %s" % (
    d["value"] / 1e3, d["ms_per_step"], r["kernel_ms"], r.get("kernel_ms_min", 0), r["all_kernels_ms"], r["frac"], d.get("bit_exact")))
for k, v in d.get("other_configs", {}).items():
    print("  %-46s %8.1f / %8.1f Gpx/s  K1 %.4f ms  frac %.4f  %s" % (k, v.get("mpix_s", 0) / 1e3, v.get("mpix_s_pipelined", 0) / 1e3, v.get("kernel_ms", 0), v.get("frac", 0), v.get("bit_exact")))
PY
[ "$MODE" = quick ] && exit 0
bash tools/pmc_phases.sh ${TAG}_struct "0 1 2 3" 2>&1 | tail -12 | tee $OUT/pmc_struct.txt
PHASE_CMD="python $PWD/tools/profile_workload.py c2noise 3" bash tools/pmc_phases.sh ${TAG}_noise "0 2 3" 2>&1 | tail -12 | tee $OUT/pmc_noise.txt
