#!/bin/bash
# SQ counters of the trellis statistics kernel (scan_segments<1, 6, 0>) for one 4K picture, per dispatch and per wave.
#   gpurun -- 'SJPEG_AMD_LIB=... bash tools/trellis_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tp1 /tmp/tp2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d /tmp/tp1 -o pmc -- python $R/tools/trellis_time.py 3840 2160 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d /tmp/tp2 -o pmc -- python $R/tools/trellis_time.py 3840 2160 3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); disp = set()
for d in ("/tmp/tp1", "/tmp/tp2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "<1, 6, 0>" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            if d.endswith("tp1"): disp.add(r["Dispatch_Id"])
n = max(len(disp), 1); waves = acc["SQ_WAVES"] / n
print("dispatches", n, "waves per dispatch %.0f" % waves)
for c, x in sorted(acc.items()):
    print("   %-24s per dispatch %.4g   per wave %.1f" % (c, x / n, x / n / waves))
PY
