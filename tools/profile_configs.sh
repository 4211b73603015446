#!/bin/bash
# rocprofv3 kernel-trace stats (and two SQ PMC passes for the configurations named in $PMC) of every
# BASELINE configuration other than the headline one (that is tools/profile_gpu.sh).
#   gpurun --timeout 1500 -- 'bash tools/profile_configs.sh r03 "c2noise c3x4 c3x1 c4 one4k m4 c5m0 c5m4" "c3x4 one4k m4"'
set -u
TAG=${1:-r03}; CONFIGS=${2:-"c2noise c3x4 c3x1 c4 one4k m4 c5m0 c5m4"}; PMC=${3:-"c3x4 one4k"}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
cd /tmp
for c in $CONFIGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -o t -- python $REPO/tools/profile_workload.py $c > $OUT/$c.log 2>&1
  for f in $(find $OUT/trace_$c -name "*kernel_stats.csv"); do cp $f $OUT/${c}_kernel_stats.csv; done
  grep "ms/step" $OUT/$c.log
  head -8 $OUT/${c}_kernel_stats.csv | cut -c1-160
done
for c in $PMC; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $OUT/pmc1_$c -o p -- python $REPO/tools/profile_workload.py $c 3 > $OUT/pmc1_$c.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc2_$c -o p -- python $REPO/tools/profile_workload.py $c 3 > $OUT/pmc2_$c.log 2>&1
  python - <<PY
import csv, collections, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for p in glob.glob("$OUT/pmc1_$c/**/*counter_collection.csv", recursive=True) + glob.glob("$OUT/pmc2_$c/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"] or 0); disp[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
with open("$OUT/${c}_pmc_summary.txt", "w") as f:
    for k, d in acc.items():
        if "scan_" not in k and "place_" not in k and "stuff_" not in k and "reduce" not in k and "adapt" not in k: continue
        f.write("== %s\n" % k)
        for cn, v in sorted(d.items()):
            n = max(len(disp[(k, cn)]), 1)
            f.write("   %-24s per_dispatch=%.4g  (dispatches %d)\n" % (cn, v / n, n))
print(open("$OUT/${c}_pmc_summary.txt").read()[:3000])
PY
done
