#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench workload.
# Results under gpurun_out/prof_<tag>/; copy the summaries you want kept into profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs"
ROOT=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
# The TIMED REGION alone: bench.py --timed-only stops after its timed regions, so the LAST
# regions x steps dispatches of every kernel in the trace ARE the timed regions (settling and warm-up calls lie
# in front of them) -- their mean is what ms_per_step is made of.
STEPS=20; REGIONS=5
rocprofv3 --kernel-trace --output-format csv -d $OUT/timed -o timed -- python $ROOT/bench.py --timed-only --steps $STEPS --regions $REGIONS > $OUT/timed.log 2>&1
python - <<PY
import csv, glob, collections
out = "$OUT"; n = $STEPS * $REGIONS
rows = []
for p in glob.glob(out + "/timed/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
line = [l for l in open(out + "/timed.log") if l.startswith("{")]
with open(out + "/timed_region_kernel_stats.csv", "w") as f:
    f.write("# the last %d dispatches per kernel of: bench.py --timed-only --steps $STEPS --regions $REGIONS (= its timed regions)\n" % n)
    f.write("kernel,dispatches,mean_us,min_us,max_us,total_ms\n")
    for k, v in sorted(by.items(), key=lambda kv: -sum(e - s for s, e in kv[1][-n:])):
        if len(v) < n or not any(t in k for t in ("scan_segments", "place_segments", "stuff_chunks", "scan_seg_offsets", "scan_chunk_offsets")):
            continue
        d = [(e - s) / 1e3 for s, e in sorted(v)[-n:]]
        f.write('"%s",%d,%.2f,%.2f,%.2f,%.3f\n' % (k, len(d), sum(d) / len(d), min(d), max(d), sum(d) / 1e3))
    if line:
        import json
        b = json.loads(line[-1])
        f.write("# bench line of the same run: ms_per_step %.4f (min %.4f max %.4f), value %.1f Mpixels/s, bit_exact %s\n" % (
            b["ms_per_step"], b["ms_per_step_min"], b["ms_per_step_max"], b["value"], b["bit_exact"]))
print(open(out + "/timed_region_kernel_stats.csv").read())
PY
PMC_CMD="python $ROOT/bench.py --timed-only --steps 3 --warmup 1 --regions 1"
pmc() {  # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/pmc_$name -o pmc -- $PMC_CMD > $OUT/pmc_$name.log 2>&1
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
