"""Host timeline (SJPEG_HIP_BATCH_DEBUG marks of sjpeg_hip_encode_batch_src) of the slow default-parameter batch calls of a process.
  python tools/slow_call_timeline.py [calls]"""
import os, re, subprocess, sys
n = sys.argv[1] if len(sys.argv) > 1 else "120"
env = dict(os.environ, SJPEG_HIP_BATCH_DEBUG="2")     # 2: also the steps of a launch that took over 200 us
p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "slow_call_hunt.py"), n], env=env, capture_output=True, text=True)
print("\n".join(l for l in p.stdout.splitlines() if l.startswith(("calls", "  #"))))
calls, cur = [], None
for l in p.stderr.splitlines():
    ms = re.match(r"\s+step (.+?)\s+([\d.]+) us", l)
    if ms and cur is not None:
        cur.append(("    -> " + ms.group(1).strip(), -9, float(ms.group(2)), cur[-1][3] if cur else 0.0)); continue
    m = re.match(r"batch (.+?)\s+part\s+(-?\d+)\s+([\d.]+) us\s+\(abs\s+([\d.]+)\)", l)
    if not m: continue
    what, part, us, ab = m.group(1).strip(), int(m.group(2)), float(m.group(3)), float(m.group(4))
    if what == "hist launched":
        cur = []; calls.append(cur)
    if cur is not None: cur.append((what, part, us, ab))
print("%d calls with a timeline" % len(calls))
prev_end = None
for i, c in enumerate(calls):
    # time between this call's first mark and the previous call's last (the part of a call in front of the marks: allocations, histogram launches)
    gap = (c[0][3] - prev_end) if prev_end is not None else 0.0
    prev_end = [x for x in c if x[1] != -9][-1][3]
    marks = [x for x in c if x[1] != -9]
    if i < 3 or marks[-1][2] > 2500 or gap > 2500:
        print("call #%d: %.0f us in front of its first mark" % (i, gap))
        for what, part, us, ab in c: print("    %-18s part %2d  %9.1f us" % (what, part, us))
