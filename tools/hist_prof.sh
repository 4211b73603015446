#!/bin/bash
# The histogram pass (tools/histogram_ablate.py: 16 x 4K frames, coefficients kept) under rocprofv3: kernel stats, then
# two counter passes (instructions / cycles, LDS).   gpurun -- 'bash tools/hist_prof.sh TAG'
set -u
export TMPDIR=/tmp
TAG=${1:-hist}
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $ROOT/tools/histogram_ablate.py 0 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $OUT/p1 -o pmc -- python $ROOT/tools/histogram_ablate.py 0 > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/p2 -o pmc -- python $ROOT/tools/histogram_ablate.py 0 > $OUT/p2.log 2>&1
python - <<PY
import csv, glob, collections
out="$OUT"
for f in glob.glob(out+"/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print("%-70s calls %5s avg %10.1f ns  %5s %%" % (r['Name'][:70], r['Calls'], float(r['AverageNs']), r['Percentage']))
acc=collections.defaultdict(lambda: collections.defaultdict(float)); disp=collections.defaultdict(set)
for f in glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value']); disp[(k,r['Counter_Name'])].add(r['Dispatch_Id'])
for k in acc:
    if 'scan_segments' not in k and 'reduce' not in k: continue
    print("==", k)
    for c,v in sorted(acc[k].items()):
        print("   %-24s per_dispatch=%.4g" % (c, v/max(len(disp[(k,c)]),1)))
PY
