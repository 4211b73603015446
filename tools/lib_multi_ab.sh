#!/bin/bash
# A/B/C... of several builds of the library on ONE box: alternates bench.py runs (K1 from HIP events, step = median
# of the timed regions).   gpurun -- 'bash tools/lib_multi_ab.sh ROUNDS tools/lib_a.bin tools/lib_b.bin ...'
# ("-" = the library in the tree)
set -u
N=$1; shift
for i in $(seq 1 "$N"); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset SJPEG_AMD_LIB; else export SJPEG_AMD_LIB=$(readlink -f "$lib"); fi
    python bench.py --no-cpu-baseline --no-other-configs --steps 20 --regions 5 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s' % '$lib', 'K1 %.4f ms (min %.4f)  step %.4f ms  ordered %.4f ms  exact %s' % (r['kernel_ms'], r['kernel_ms_min'], d['ms_per_step'], d['ms_per_step_ordered'], d['bit_exact']))"
  done
done
