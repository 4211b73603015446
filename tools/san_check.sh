#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer run of the HOST code (host_api.cc, jpeg_host.cc,
# jpeg_tools.cc: buffer, stream and cache management of the drop-in API) under the C++ API test:
# the three files are compiled with g++ -fsanitize=address,undefined and linked with the ordinary
# device objects into a scratch copy of the library; tests/cxx/api_test.cc (sanitized too) runs its
# whole programme and the out-of-the-box AUTO calls against it on the GPU.
#   gpurun --timeout 900 -- 'bash tools/san_check.sh'        (run `make -C sjpeg_amd/csrc` first: device objects)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/sjpeg_san}
rm -rf "$W"; mkdir -p "$W/out" "$W/out2"
C=$ROOT/sjpeg_amd/csrc
# (-fno-sanitize=enum: the API test hands SjpegEncode an out-of-range SjpegYUVMode on purpose -- the argument
# check under test -- which is a finding in the CALLER by the letter of the language)
SAN="-fsanitize=address,undefined -fno-sanitize=enum -fno-omit-frame-pointer -g -O1"
for f in host_api jpeg_host jpeg_tools; do
  g++ -std=c++17 -fPIC -ffp-contract=off $SAN -I"$ROOT/include" -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -c "$C/$f.cc" -o "$W/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic "$C/scan_engine.o" "$C/sharp_yuv.o" "$C/riskiness.o" "$C/exchange.o" \
    "$W/host_api.o" "$W/jpeg_host.o" "$W/jpeg_tools.o" -ldl -o "$W/libsjpeg_amd.so"
cp "$C/riskiness.bin" "$W/"
g++ -std=c++17 $SAN -I"$ROOT/include" "$ROOT/tests/cxx/api_test.cc" -o "$W/api_test" -L"$W" -lsjpeg_amd -lpthread \
    -Wl,-rpath,"$W" -Wl,-rpath-link,/opt/rocm/lib
export LD_LIBRARY_PATH=/opt/rocm/lib:${LD_LIBRARY_PATH:-}
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== api_test (whole programme)"
set +e; "$W/api_test" "$W/out" > "$W/run1.log" 2>&1; echo "exit code $?"; grep -m3 -A12 "ERROR\|runtime error" "$W/run1.log"; tail -3 "$W/run1.log"
echo "== api_test --auto (SjpegCompress + default EncoderParam)"
"$W/api_test" "$W/out2" --auto "$ROOT/tests/golden/test128.rgb" 128 128 > "$W/run2.log" 2>&1; echo "exit code $?"; grep -m3 -A12 "ERROR\|runtime error" "$W/run2.log"; tail -3 "$W/run2.log"
md5sum "$W/out2/compress_c1.jpg" | cut -c1-32
