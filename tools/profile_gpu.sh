#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench workload.
# Results under gpurun_out/prof_<tag>/; copy the summaries you want kept into profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
pmc() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
  for f in $(find $OUT/pmc_$name -name "*counter_collection.csv"); do cp $f $OUT/pmc_$name.csv; done
}
pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc grbm GRBM_GUI_ACTIVE
python - <<PY
import csv, collections, glob, os
out = "$OUT"
def agg(path):
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r.get("Kernel_Name") or r.get("Kernel Name")
        c = r.get("Counter_Name"); v = float(r.get("Counter_Value") or 0)
        acc[k][c] += v
    return acc
summary = {}
for p in sorted(glob.glob(out + "/pmc_*.csv")):
    for k, d in agg(p).items():
        summary.setdefault(k, {}).update(d)
disp = collections.Counter()
for p in glob.glob(out + "/pmc_sq1.csv"):
    seen = set()
    for r in csv.DictReader(open(p)):
        key = (r.get("Dispatch_Id"), r.get("Kernel_Name"))
        if key not in seen:
            seen.add(key); disp[r.get("Kernel_Name")] += 1
with open(out + "/pmc_summary.txt", "w") as f:
    for k, d in summary.items():
        n = max(disp.get(k, 1), 1)
        f.write(f"== {k}  dispatches={n}\n")
        for c, v in sorted(d.items()):
            f.write(f"   {c:28s} total={v:.4g}  per_dispatch={v/n:.4g}\n")
print(open(out + "/pmc_summary.txt").read()[:6000])
PY
head -20 $OUT/kernel_stats.csv
