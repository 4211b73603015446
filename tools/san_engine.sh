#!/bin/bash
# AddressSanitizer on the HOST half of scan_engine.hip (the engine: buffers, streams, child engines and the state machine of
# the batch path's lanes, uploads through pinned blocks) -- tools/san_check.sh covers the .cc files of the drop-in API only.
# The device code is compiled as usual (-fno-gpu-sanitize).  Built here (hipcc cross-compiles), run on the GPU box:
#   tools/san_engine.sh build && gpurun --timeout 900 -- 'bash tools/san_engine.sh run'
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/sjpeg_amd/csrc
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
if [ "${1:-build}" = build ]; then
  make -s -C "$C"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fsanitize=address -shared-libsan -fno-gpu-sanitize \
      -fno-omit-frame-pointer -I"$ROOT/include" -c "$C/scan_engine.hip" -o /tmp/scan_engine_asan.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -shared-libsan -Wl,-Bsymbolic /tmp/scan_engine_asan.o "$C/sharp_yuv.o" "$C/riskiness.o" \
      "$C/exchange.o" "$C/host_api.o" "$C/jpeg_host.o" "$C/jpeg_tools.o" -ldl -o "$ROOT/tools/lib_asan.bin"
  SAN="-fsanitize=address -shared-libasan -fno-omit-frame-pointer -g -O1"
  /opt/rocm/lib/llvm/bin/clang++ -std=c++17 $SAN -I"$ROOT/include" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ "$ROOT/tests/cxx/batch_lanes_test.cc" \
      -o "$ROOT/tools/batch_lanes_test_asan.bin" "$ROOT/tools/lib_asan.bin" -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
  g++ -std=c++17 -O1 -I"$ROOT/include" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ "$ROOT/tests/cxx/batch_lanes_test.cc" \
      -o "$ROOT/tools/batch_lanes_test.bin" -L"$C" -lsjpeg_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,"$C" -Wl,-rpath,/opt/rocm/lib
  echo "tools/lib_asan.bin tools/batch_lanes_test_asan.bin tools/batch_lanes_test.bin"
  exit 0
fi
if [ "${1:-}" = run ]; then
  # the C++ driver (no Python, no torch: AddressSanitizer and their copy of the HIP runtime do not get along)
  export LD_LIBRARY_PATH=/opt/rocm/lib:$(dirname "$RT"):${LD_LIBRARY_PATH:-}
  export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0
  cp "$C/riskiness.bin" "$ROOT/tools/" 2>/dev/null || true
  for env in "SJPEG_HIP_BATCH_JOB_MPIX=0.02" "SJPEG_HIP_BATCH_JOB_MPIX=0.3 SJPEG_HIP_BATCH_LANES=2" "SJPEG_HIP_BATCH_LANES=0" "X=1"; do
    echo "== plain build, $env"; env $env "$ROOT/tools/batch_lanes_test.bin" 12 2>&1 | grep -v amdgpu.ids | tail -2
    echo "== engine under AddressSanitizer, $env"
    set +e
    env $env "$ROOT/tools/batch_lanes_test_asan.bin" 12 > /tmp/asan_run.log 2>&1; echo "exit code $?"
    grep -m2 -A18 "ERROR: AddressSanitizer" /tmp/asan_run.log; grep -v amdgpu.ids /tmp/asan_run.log | tail -2
    set -e
  done
  exit 0
fi
export SJPEG_AMD_LIB=$ROOT/tools/lib_asan.bin
export LD_PRELOAD=$RT
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0:allocator_may_return_null=1
cd "$ROOT"
run() { echo "== $*"; "$@" 2>&1 | grep -v "amdgpu.ids" | grep -m3 -A14 "ERROR: AddressSanitizer" ; echo "exit ${PIPESTATUS[0]}"; }
echo "== lanes_check (23 one-frame jobs on four lanes)"; SJPEG_HIP_BATCH_JOB_MPIX=0.02 python tests/lanes_check.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "== lanes_check (default: one job)"; python tests/lanes_check.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "== lanes_check (round 5's two parts)"; SJPEG_HIP_BATCH_LANES=0 python tests/lanes_check.py 2>&1 | grep -v amdgpu.ids | tail -3
echo "== batch fuzz, lanes forced, 40 s"; SJPEG_HIP_BATCH_JOB_MPIX=0.05 python tools/batch_fuzz.py 7 40 2>&1 | grep -v amdgpu.ids | tail -3
echo "== engine churn"; python tools/engine_churn.py 20 2>&1 | grep -v amdgpu.ids | tail -2
echo "== GPU tests on the sanitized engine (batch / exchange / band / trim subsets)"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or exchange or band or trim or scratch or lanes or methods or c4 or stride" 2>&1 | grep -v "amdgpu.ids\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
