#!/bin/bash
# Disassembly + resource usage of K1 (scan_segments<MODE, KIND, SRC>, default <1, 0, 0>) out of the built library.
#   tools/k1_isa.sh [out_dir] ["1, 0, 0"]     -> out_dir/k1.s, prints VGPRs / scratch / LDS and instruction counts
set -eu
OUT=${1:-/tmp/isa}; INST=${2:-1, 0, 0}
LIB=$(cd "$(dirname "$0")/.." && pwd)/sjpeg_amd/csrc/libsjpeg_amd.so
LLVM=/opt/rocm/lib/llvm/bin
rm -rf "$OUT"; mkdir -p "$OUT"; cp "$LIB" "$OUT/lib.so"; cd "$OUT"
$LLVM/llvm-objdump --offloading lib.so > /dev/null
CO=$(ls -S lib.so.*.hipv4-* | head -1)
$LLVM/llvm-objdump -d --no-show-raw-insn "$CO" | c++filt > all.s
awk -v pat="scan_segments<$INST>" 'index($0, pat) && /^[0-9a-f]+ </ {on=1; print; next} on && /^[0-9a-f]+ </ {exit} on {print}' all.s | sed 's#//.*##' > k1.s
$LLVM/llvm-readelf --notes "$CO" | c++filt | awk -v pat="scan_segments<$INST>" '/\.name:/ {on = index($0, pat) > 0} on && /vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill/ {print}' | sort -u
echo "lines $(wc -l < k1.s)  VALU $(grep -c '^\s*v_' k1.s)  SALU $(grep -c '^\s*s_' k1.s)  LDS $(grep -c '^\s*ds_' k1.s)"
