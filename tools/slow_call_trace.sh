#!/bin/bash
# Names the slow call of the default-parameter batch path: HIP API + kernel + copy trace of N calls,
# then every HIP API call, kernel and copy over 1 ms outside the first two calls (which allocate the scratch).
#   gpurun -- 'bash tools/slow_call_trace.sh [calls]'
set -u
N=${1:-300}
OUT=$PWD/gpurun_out/r05/slowcall; mkdir -p $OUT
ROOT=$PWD
export TMPDIR=/tmp
cd /tmp
for run in 1 2 3; do
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $OUT/t$run -o t -- python $ROOT/tools/slow_call_hunt.py $N > $OUT/hunt$run.log 2>&1
grep -A8 "^calls" $OUT/hunt$run.log
python - <<PY
import csv, glob
out = "$OUT/t$run"
def rows(pat):
    r = []
    for p in glob.glob(out + "/**/*" + pat, recursive=True): r += list(csv.DictReader(open(p)))
    return r
api = rows("hip_api_trace.csv"); ker = rows("kernel_trace.csv"); cp = rows("memory_copy_trace.csv")
api.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(api[0]["Start_Timestamp"])
# the batch calls: every call launches scan_segments<1,2,0> (histogram kind) twice; find the time of the 3rd call's start
hist = sorted(int(k["Start_Timestamp"]) for k in ker if "scan_segments<1, 2, 0>" in k["Kernel_Name"])
cut = hist[4] if len(hist) > 4 else t0
print("run $run: %d API calls, %d kernels, %d copies; reporting everything over 1 ms after the second batch call" % (len(api), len(ker), len(cp)))
for r in api:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 1.0 and int(r["Start_Timestamp"]) > cut:
        print("  API  %-32s %8.3f ms at %10.3f ms" % (r["Function"], d, (int(r["Start_Timestamp"]) - t0) / 1e6))
for r in ker:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 1.0 and int(r["Start_Timestamp"]) > cut:
        print("  KERNEL %-40s %8.3f ms at %10.3f ms" % (r["Kernel_Name"][:40], d, (int(r["Start_Timestamp"]) - t0) / 1e6))
for r in cp:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 1.0 and int(r["Start_Timestamp"]) > cut:
        print("  COPY %-20s %8.3f ms at %10.3f ms" % (r.get("Direction", "?"), d, (int(r["Start_Timestamp"]) - t0) / 1e6))
# gaps between consecutive histogram kernels (a call every ~1.3 ms): where the timeline stretched
g = [(hist[i + 1] - hist[i]) / 1e6 for i in range(len(hist) - 1)]
big = [(i, x) for i, x in enumerate(g) if x > 2.5 and i > 4]
print("  gaps over 2.5 ms between histogram launches (index, ms):", big[:10])
PY
rm -rf $OUT/t$run
done
