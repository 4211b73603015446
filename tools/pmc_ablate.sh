#!/bin/bash
# PMC comparison across ablation levels (GPU box). Usage: tools/pmc_ablate.sh "2 3 0"
set -u
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_ablate
mkdir -p $OUT
cd /tmp
for A in $1; do
  G=0
  for P in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_IFETCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_LEVEL_WAVES SQ_IFETCH_LEVEL SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM"; do
    G=$((G+1)); tag=g$G
    SJPEG_HIP_ABLATE=$A rocprofv3 --kernel-trace --output-format csv --pmc $P -d $OUT/a${A}_$tag -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a${A}_$tag.log 2>&1
  done
done
python - <<PY
import csv, glob, collections, os
out="$OUT"
res=collections.defaultdict(dict)
for d in sorted(glob.glob(out+"/a*_*")):
    if not os.path.isdir(d): continue
    a=os.path.basename(d).split('_')[0]
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); disp=set()
        for r in csv.DictReader(open(f)):
            if 'scan_segments' not in r['Kernel_Name']: continue
            acc[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
        for k,v in acc.items(): res[a][k]=v/max(len(disp),1)
keys=sorted({k for a in res for k in res[a]})
levels=sorted(res)
print("counter".ljust(26)+"".join(l.rjust(14) for l in levels))
for k in keys:
    print(k.ljust(26)+"".join(("%.4g"%res[l].get(k,float('nan'))).rjust(14) for l in levels))
PY
