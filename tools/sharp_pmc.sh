#!/bin/bash
# SQ counters of sharp_sweeps_strips for one 1080p picture (host API), per dispatch and per wave.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sp1 /tmp/sp2
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD -d /tmp/sp1 -o pmc -- python $R/tools/sharp_time.py ${1:-1920} ${2:-1080} 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA -d /tmp/sp2 -o pmc -- python $R/tools/sharp_time.py ${1:-1920} ${2:-1080} 3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); disp = set()
for d in ("/tmp/sp1", "/tmp/sp2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sharp_sweeps_strips" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            if d.endswith("sp1"): disp.add(r["Dispatch_Id"])
n = max(len(disp), 1); waves = acc["SQ_WAVES"] / n
print("dispatches", n, "waves per dispatch %.0f" % waves)
for c, x in sorted(acc.items()):
    print("   %-24s per dispatch %.4g   per wave %.1f" % (c, x / n, x / n / waves))
PY
