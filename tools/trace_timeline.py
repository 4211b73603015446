"""Timeline of the last kernels of a rocprofv3 --kernel-trace run (csv): start / end relative to the first one shown,
queue, duration, gap to the end of the previous kernel.  Usage: python tools/trace_timeline.py <dir> [n_last]"""
import csv
import glob
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for p in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
last_end = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:44]
    print("%9.1f %9.1f  dur %7.1f  idle %7.1f  q%-3s %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, max(s - last_end, 0) / 1e3, r.get("Queue_Id", "?"), name))
    last_end = max(last_end, e)
