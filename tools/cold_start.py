"""Cold-start harness for BASELINE config C1: N fresh PROCESSES (not loops), each making the reference's
two out-of-the-box calls -- SjpegCompress() and sjpeg::Encode(default EncoderParam), both
SJPEG_YUV_AUTO -> riskiness scan, sharp-YUV conversion, method 4 -- as its FIRST calls into the
library (tests/cxx/api_test.cc --auto), P of them at a time.  Every process pays the first module
load, the first discovery of riskiness.bin, the first table uploads and a cold per-thread context.
Any non-zero exit or wrong byte is kept: stdout / stderr / the output files go to
<logdir>/fail_<i>/.  Usage (GPU box, repo root): python tools/cold_start.py N [P] [logdir]"""
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
LOG = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "gpurun_out", "r03", "cold_start")
WANT = "acc8ce8111f5ff4b32b3faa15ad5d994"
CSRC = os.path.join(ROOT, "sjpeg_amd", "csrc")
RGB = os.path.join(ROOT, "tests", "golden", "test128.rgb")

os.makedirs(LOG, exist_ok=True)
work = tempfile.mkdtemp(prefix="cold_")
exe = os.path.join(work, "api_test")
subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tests", "cxx", "api_test.cc"), "-o", exe, "-L", CSRC, "-lsjpeg_amd",
                       "-lpthread", "-Wl,-rpath," + CSRC, "-Wl,-rpath-link,/opt/rocm/lib"])
env = dict(os.environ)
env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
env.pop("SJPEG_HIP_RISKINESS_TABLE", None)
TABLE = os.path.join(CSRC, "riskiness.bin")


def one(i):
    out = os.path.join(work, "o%d" % i)
    os.makedirs(out)
    e = dict(env)
    if i % 3 == 2:                                  # every third process finds the table through the environment
        e["SJPEG_HIP_RISKINESS_TABLE"] = TABLE
    r = subprocess.run([exe, out, "--auto", RGB, "128", "128"], capture_output=True, text=True, env=e)
    why = None
    if r.returncode != 0:
        why = "exit code %d" % r.returncode
    else:
        for name in ("compress_c1.jpg", "default_param_auto.jpg"):
            path = os.path.join(out, name)
            data = open(path, "rb").read() if os.path.exists(path) else b""
            if hashlib.md5(data).hexdigest() != WANT:
                why = "%s: %d bytes, md5 %s" % (name, len(data), hashlib.md5(data).hexdigest())
    if why is not None:
        keep = os.path.join(LOG, "fail_%d" % i)
        shutil.copytree(out, keep, dirs_exist_ok=True)
        with open(os.path.join(keep, "why.txt"), "w") as f:
            f.write(why + "\n--- stdout\n" + r.stdout + "\n--- stderr\n" + r.stderr)
    shutil.rmtree(out, ignore_errors=True)
    return why


t0 = time.time()
bad = []
with cf.ThreadPoolExecutor(P) as ex:
    for i, why in enumerate(ex.map(one, range(N))):
        if why is not None:
            bad.append((i, why))
            print("FAIL process %d: %s" % (i, why), flush=True)
shutil.rmtree(work, ignore_errors=True)
line = "cold start: %d fresh processes (%d at a time), %d failures, %.0f s" % (N, P, len(bad), time.time() - t0)
print(line)
with open(os.path.join(LOG, "summary.txt"), "a") as f:
    f.write(line + "\n")
    for i, why in bad:
        f.write("  process %d: %s\n" % (i, why))
