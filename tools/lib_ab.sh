#!/bin/bash
# A/B of two builds of the library on ONE box: alternates bench.py runs (K1 time from HIP events).
#   gpurun -- 'bash tools/lib_ab.sh tools/lib_old.bin [rounds]'
set -u
OLD=$(readlink -f "$1"); N=${2:-3}
for i in $(seq 1 "$N"); do
  for which in old new; do
    if [ $which = old ]; then export SJPEG_AMD_LIB=$OLD; else unset SJPEG_AMD_LIB; fi
    python bench.py --no-cpu-baseline --no-other-configs --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$which', 'K1 %.4f ms  step %.4f ms  ordered %.4f ms  exact %s' % (r['kernel_ms'], d['ms_per_step'], d['ms_per_step_ordered'], d['bit_exact']))"
  done
done
