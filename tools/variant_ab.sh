#!/bin/bash
# Variant builds side by side on ONE box: K1 time (bench.py, HIP events), VALU per wave of K1 (struct and noise), parity.
#   gpurun -- 'bash tools/variant_ab.sh ROUNDS - tools/lib_a.bin ...'      ("-" = the library in the tree)
set -u
N=$1; shift
bash tools/lib_multi_ab.sh "$N" "$@"
for lib in "$@"; do
  if [ "$lib" = "-" ]; then L=""; T=tree; else L=$lib; T=$(basename $lib .bin); fi
  echo "== $T"
  bash tools/pmc_phases.sh ab_${T}_s "0" $L 2>&1 | grep "VALU/wave\|SQ_INSTS_SALU\|SQ_INSTS_LDS"
  PHASE_CMD="python $PWD/tools/profile_workload.py c2noise 3" bash tools/pmc_phases.sh ab_${T}_n "0" $L 2>&1 | grep "VALU/wave"
  if [ -n "$L" ]; then SJPEG_AMD_LIB=$(readlink -f $L) python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_sizes or c2_4k or extreme or segments_of_every" 2>&1 | tail -1; fi
done
