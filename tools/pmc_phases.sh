#!/bin/bash
# VALU / SALU / LDS instructions and wave cycles of K1 per ablation level (GPU box).
#   tools/pmc_phases.sh TAG "1 2 3 0" [lib.so]      (PHASE_CMD="python tools/profile_workload.py c3x4 3": another workload)
set -u
export TMPDIR=/tmp
TAG=$1; LEVELS=$2
[ -n "${3:-}" ] && export SJPEG_AMD_LIB=$PWD/$3
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
ROOT=$PWD
cd /tmp
for A in $LEVELS; do
  SJPEG_HIP_ABLATE=$A rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $OUT/a$A -o pmc -- ${PHASE_CMD:-python $ROOT/bench.py --steps 3 --warmup 1 --regions 1 --timed-only ${BENCH_EXTRA:-}} > $OUT/a$A.log 2>&1
done
python - <<PY
import csv, glob, collections, os
out="$OUT"
res=collections.defaultdict(dict)
for d in sorted(glob.glob(out+"/a*")):
    if not os.path.isdir(d): continue
    a=os.path.basename(d)
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); disp=set()
        for r in csv.DictReader(open(f)):
            if 'scan_segments' not in r['Kernel_Name']: continue
            acc[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
        for k,v in acc.items(): res[a][k]=v/max(len(disp),1)
keys=sorted({k for a in res for k in res[a]})
levels=sorted(res)
lines=["counter".ljust(22)+"".join(l.rjust(12) for l in levels)]
for k in keys:
    lines.append(k.ljust(22)+"".join(("%.4g"%res[l].get(k,float('nan'))).rjust(12) for l in levels))
lines.append("VALU/wave".ljust(22)+"".join(("%.0f"%(res[l].get('SQ_INSTS_VALU',0)/max(res[l].get('SQ_WAVES',1),1))).rjust(12) for l in levels))
open(out+"/summary.txt","w").write("\n".join(lines)+"\n")
print("\n".join(lines))
PY
