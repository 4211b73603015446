#!/bin/bash
# Everything that guards parity, on a GPU box (through gpurun: /usr/local/graft/bin/gpurun --timeout 1800 -- 'bash tools/run_all_checks.sh').
# Builds nothing itself except the stress configuration, and ALWAYS leaves the plain (shipped) build behind.
set -u
cd "$(dirname "$0")/.."
run() { echo "== $*"; timeout "${T:-600}" "$@" 2>&1 | grep -v "amdgpu.ids\|^RCCL version\|^HIP version\|^ROCm version\|^Hostname\|^Librccl path" | tail -${N:-3}; }
N=2 run python -m pytest tests -m gpu -x -q
N=1 run python tools/gpu_fuzz.py 1 240
N=3 run python tools/low_entropy_fuzz.py 1 "${LOWENT:-120}"
N=3 run python tools/extremes_fuzz.py 1 "${LOWENT:-120}"
N=1 run python tools/trellis_fuzz.py "${TRELLIS:-300}"
N=1 run python tools/batch_fuzz.py 1 "${LOWENT:-120}"
# the same fuzz with the batch path's LANES forced onto these small pictures (round 6: jobs of 0.02 / 0.3 Mpixels on four / two lanes)
N=1 run env SJPEG_HIP_BATCH_JOB_MPIX=0.02 python tools/batch_fuzz.py 2 "${LOWENT:-120}"
N=1 run env SJPEG_HIP_BATCH_JOB_MPIX=0.3 SJPEG_HIP_BATCH_LANES=2 python tools/batch_fuzz.py 3 "${LOWENT:-120}"
N=1 run python tests/lanes_check.py
N=1 run python tools/gpu_soak.py 1 "${SOAK:-120}"
N=1 run python tools/sharp_soak.py "${SOAK:-60}" 1
N=1 run python tools/thread_soak.py 16 100
N=1 run python tools/sharp_threads.py 8 60
N=2 run python tools/engine_churn.py 100
N=5 run python tools/many_frames_check.py
if command -v hipcc >/dev/null 2>&1 || [ -x /opt/rocm/bin/hipcc ]; then
  for s in 1 2; do
    make -s -C sjpeg_amd/csrc STRESS=$s || exit 1
    echo "== stress build $s"
    N=2 run python -m pytest tests -m gpu -x -q
    N=2 run python tools/race_sweep.py
    N=1 run python tools/gpu_soak.py $((40 + s)) 60
  done
  make -s -C sjpeg_amd/csrc || exit 1
  ls -a sjpeg_amd/csrc | grep config_
fi
N=1 run python bench.py --no-cpu-baseline
