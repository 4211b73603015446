#!/bin/bash
# A/B of one environment setting on ONE box: alternates bench.py runs.
#   gpurun -- 'bash tools/env_ab.sh NAME VALUE_A VALUE_B [rounds]'   ("-" = unset)
set -u
NAME=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq 1 "$N"); do
  for v in "$A" "$B"; do
    if [ "$v" = "-" ]; then unset "$NAME"; else export "$NAME=$v"; fi
    python bench.py --no-cpu-baseline --no-other-configs --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$NAME=$v', 'K1 %.4f ms  step %.4f ms  ordered %.4f ms  exact %s' % (r['kernel_ms'], d['ms_per_step'], d['ms_per_step_ordered'], d['bit_exact']))"
  done
done
