"""Builds tests/cxx/api_test.cc against include/sjpeg.h + libsjpeg_amd.so (the way a user of the
reference would build against sjpeg.h + libsjpeg) and runs it on the GPU; its saved outputs are
compared byte for byte with the oracle."""
import os
import subprocess

import numpy as np
import pytest

import sjpeg_amd as sj
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cxx", "api_test.cc")


def _build(tmpdir, libdir=None):
    libdir = libdir or sj.CSRC
    exe = os.path.join(tmpdir, "api_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC,
                           "-o", exe, "-L", libdir, "-lsjpeg_amd", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def test_cxx_api_compiles_and_links(tmp_path):
    """CPU: the public header is self-contained C++ and every symbol it promises links."""
    _build(str(tmp_path))


@pytest.mark.gpu
def test_cxx_api_behaviour_and_parity(tmp_path, oracle):
    exe = _build(str(tmp_path))
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    img = synth.g_struct(141, 99, 4242)
    modes = {"420": 1, "444": 3, "400": 4}

    def read(name):
        return (out / (name + ".jpg")).read_bytes()

    q75 = sj.make_tables(quality=75.0)[1]
    q33 = sj.make_tables(quality=33.0)[1]
    for name, mode in modes.items():
        assert read("default_" + name) == oracle.encode_full(img, q75, yuv_mode=mode, method=4), name
        assert read("q33_adaptive_bias60_d7_3_" + name) == oracle.encode_full(
            img, q33, q_bias=0x60, dmax_luma=7, dmax_chroma=3, yuv_mode=mode, method=3), name
    m = np.array([[3 + i for i in range(64)], [5 + 2 * i for i in range(64)]], np.float64)
    quant = np.clip((m * 100.0 / 80.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)
    assert read("setquant_r80_limit_420") == oracle.encode_full(img, quant, min_quant=quant, yuv_mode=1, method=4)
    assert read("threads_q72_420") == oracle.encode_full(img, sj.make_tables(quality=72.0)[1], yuv_mode=1, method=4)
    # other input layouts, rebuilt here exactly as api_test.cc builds them
    w, h = 37, 23
    cw, ch = (w + 1) // 2, (h + 1) // 2
    px = synth.g_struct(w, h, 777).reshape(-1)
    rgb = px.reshape(h * w, 3)
    bgra = np.stack([rgb[:, 2], rgb[:, 1], rgb[:, 0], np.full(h * w, 0x5a, np.uint8)], 1).reshape(h, 4 * w)
    yp, u4, v4 = (rgb[:, c].reshape(h, w).copy() for c in range(3))
    idx = np.arange(cw * ch)
    up = px[(5 * idx) % px.size].reshape(ch, cw).copy()
    vp = px[(7 * idx) % px.size].reshape(ch, cw).copy()
    uv = np.stack([up.reshape(-1), vp.reshape(-1)], 1).reshape(ch, 2 * cw).copy()
    q66 = sj.make_tables(quality=66.0)[1]
    from oracle import orc
    assert read("bgra_444_q66") == oracle.encode_src(orc.SRC_BGRA, [bgra], w, h, q66, yuv_mode=3, method=4)
    assert read("gray_q66") == oracle.encode_src(orc.SRC_GRAY, [yp], w, h, q66, method=4)
    assert read("nv12_q66") == oracle.encode_src(orc.SRC_NV12, [yp, uv], w, h, q66, method=4)
    assert read("nv21_q66") == oracle.encode_src(orc.SRC_NV21, [yp, uv], w, h, q66, method=4)
    assert read("yuv420_q66") == oracle.encode_src(orc.SRC_YUV420, [yp, up, vp], w, h, q66, method=4)
    assert read("yuv444_q66") == oracle.encode_src(orc.SRC_YUV444, [yp, u4, v4], w, h, q66, method=4)
    # metadata: same entropy data and tables as the plain stream, with the APPn segments spliced in
    meta = read("metadata_444")
    plain = oracle.encode_full(img, sj.make_tables(quality=80.0)[1], yuv_mode=3, method=0)
    assert meta[:20] == plain[:20] and meta.endswith(plain[20:])
    extra = meta[20:len(meta) - (len(plain) - 20)]
    assert extra.startswith(b"\xff\xe5\x00\x04zz" + b"\xff\xe1") and b"\xff\xe1\x00\x1cExif\x00\x00II*\x00fake-exif-payloa\xff\xe2\xff\xffICC_PROFILE\x00\x01\x02" in extra
    assert extra.count(b"ICC_PROFILE\x00") == 2 and b"http://ns.adobe.com/xap/1.0/\x00<x:xmpmeta/>" in extra
    assert read("trellis_q70_420") == oracle.encode_full(img, sj.make_tables(quality=70.0)[1], yuv_mode=1, method=7)
    # multi-pass searches: same nested loops as api_test.cc
    q60 = sj.make_tables(quality=60.0)[1]
    idx = 0
    for name, mode in modes.items():
        for huff in (False, True):
            for adapt in (False, True):
                for tm in (1, 2):
                    for t in range(3):
                        for passes in (2, 6):
                            target = (1500.0, 4000.0, 9000.0)[t] if tm == 1 else (30.0, 38.0, 45.0)[t]
                            want = oracle.encode_search(orc.SRC_RGB, [img], 141, 99, q60, yuv_mode=mode,
                                                        huffman=huff, adaptive=adapt, target_mode=tm,
                                                        target_value=target, passes=passes,
                                                        tolerance=1.0 if tm == 1 else 0.1)
                            assert read("search_%03d" % idx) == want, (idx, name, huff, adapt, tm, target, passes)
                            idx += 1
    for name, mode in modes.items():
        for tm in (1, 2):
            for t in range(3):
                for passes in (2, 6):
                    target = (1500.0, 4000.0, 9000.0)[t] if tm == 1 else (30.0, 38.0, 45.0)[t]
                    want = oracle.encode_search(orc.SRC_RGB, [img], 141, 99, q60, yuv_mode=mode, target_mode=tm,
                                                target_value=target, passes=passes,
                                                tolerance=1.0 if tm == 1 else 0.1, trellis=True)
                    assert read("search_trellis_%03d" % idx) == want, (idx, name, tm, target, passes)
                    idx += 1
    assert read("search_hooked") == oracle.encode_search(orc.SRC_RGB, [img], 141, 99, q60, yuv_mode=1,
                                                         target_mode=1, target_value=5000.0, passes=4)


@pytest.mark.gpu
def test_cxx_out_of_the_box_auto_mode(tmp_path):
    """SjpegCompress() and sjpeg::Encode(default EncoderParam) -- the reference's out-of-the-box calls,
    both SJPEG_YUV_AUTO -- through a C++ caller that installs NOTHING: the library finds the riskiness
    table by itself (file next to it, or the file named by SJPEG_HIP_RISKINESS_TABLE), and without any it
    fails loudly instead of picking a colour mode.  BASELINE config #1's known answer."""
    import hashlib
    import shutil
    table = os.path.join(sj.CSRC, "riskiness.bin")
    if not os.path.exists(table):
        pytest.skip("sjpeg_amd/csrc/riskiness.bin not installed (__graft_entry__.build() extracts it from oracle/_ref)")
    rgb = os.path.join(ROOT, "tests", "golden", "test128.rgb")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    env.pop("SJPEG_HIP_RISKINESS_TABLE", None)

    def run(exe, outdir, extra_env=None, expect_fail=False):
        os.makedirs(outdir, exist_ok=True)
        e = dict(env)
        e.update(extra_env or {})
        args = [exe, outdir, "--auto", rgb, "128", "128"] + (["expect-fail"] if expect_fail else [])
        r = subprocess.run(args, capture_output=True, text=True, env=e)
        assert r.returncode == 0, r.stdout + r.stderr
        if not expect_fail:
            got = open(os.path.join(outdir, "compress_c1.jpg"), "rb").read()
            assert len(got) == 2571 and hashlib.md5(got).hexdigest() == "acc8ce8111f5ff4b32b3faa15ad5d994"
            assert open(os.path.join(outdir, "default_param_auto.jpg"), "rb").read() == got   # same call by another name

    # 1. the shipped layout: riskiness.bin next to libsjpeg_amd.so
    run(_build(str(tmp_path)), str(tmp_path / "o1"))
    # 2. a copy of the library alone in another directory: nothing to find -> loud failure;
    #    then the environment variable; then the file next to THAT copy
    libdir = tmp_path / "lib"
    libdir.mkdir()
    shutil.copy(sj.LIB_PATH, libdir / "libsjpeg_amd.so")
    exe2dir = tmp_path / "exe2"
    exe2dir.mkdir()
    exe2 = _build(str(exe2dir), str(libdir))
    run(exe2, str(tmp_path / "o2"), expect_fail=True)
    run(exe2, str(tmp_path / "o3"), {"SJPEG_HIP_RISKINESS_TABLE": table})
    shutil.copy(table, libdir / "riskiness.bin")
    run(exe2, str(tmp_path / "o4"))


@pytest.mark.gpu
def test_cxx_batch_entry_against_the_host_api(tmp_path):
    """sjpeg_hip_encode_batch_src from a plain C++ process (tests/cxx/batch_lanes_test.cc: hipMalloc'd frames, no torch):
    every frame of every batch equals SjpegEncode's single-picture encode of it -- with the batch cut into one-frame jobs on
    four lanes, into two lanes, into round 5's two parts, and by default.  (tools/san_engine.sh runs the same programme
    against an engine compiled with AddressSanitizer: profiles/r06/san_engine.txt.)"""
    exe = os.path.join(str(tmp_path), "batch_lanes_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "cxx", "batch_lanes_test.cc"), "-o", exe,
                           "-L", sj.CSRC, "-lsjpeg_amd", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + sj.CSRC, "-Wl,-rpath,/opt/rocm/lib"])
    for extra in ({"SJPEG_HIP_BATCH_JOB_MPIX": "0.02"}, {"SJPEG_HIP_BATCH_JOB_MPIX": "0.3", "SJPEG_HIP_BATCH_LANES": "2"},
                  {"SJPEG_HIP_BATCH_LANES": "0"}, {}):
        env = dict(os.environ)
        env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
        env.update(extra)
        r = subprocess.run([exe, "8"], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and "mismatches: 0" in r.stdout, (extra, r.stdout[-500:], r.stderr[-1500:])
