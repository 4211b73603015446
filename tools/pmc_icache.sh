#!/bin/bash
# Instruction-cache behaviour of K1 on the bench workload (GPU box): requests, hits, misses, fetch stalls.
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05/icache; mkdir -p $OUT; ROOT=$PWD
[ -n "${1:-}" ] && export SJPEG_AMD_LIB=$PWD/$1
cd /tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/$n -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --regions 1 --timed-only > $OUT/$n.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_segments" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc): print("%-32s %.4g per launch" % (k, acc[k] / max(len(disp[k]), 1)))
PY
