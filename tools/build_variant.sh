#!/bin/bash
# A variant build of the library for A/B runs (tools/lib_multi_ab.sh): scan_engine.hip compiled with extra flags,
# linked with the tree's other objects into tools/lib_NAME.bin (git-ignored, travels to the GPU box).
#   tools/build_variant.sh NAME "-DSJPEG_WALK_PIPE=1"
set -eu
NAME=$1; FLAGS=${2:-}
cd "$(dirname "$0")/../sjpeg_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -I../../include $FLAGS -c scan_engine.hip -o /tmp/scan_engine_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic /tmp/scan_engine_$NAME.o sharp_yuv.o riskiness.o exchange.o host_api.o jpeg_host.o jpeg_tools.o -ldl -o ../../tools/lib_$NAME.bin
echo "tools/lib_$NAME.bin"
