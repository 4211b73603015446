"""Parity tests proper: the HIP path, called through the C-ABI (ctypes), against the oracle on
the same seeded inputs, against the committed golden vectors, and at BASELINE.json's full
sizes through digests + size-independent properties.  Bar: BIT-EXACT (integer/byte work)."""
import ctypes as C
import hashlib
import io
import os
import threading

import numpy as np
import pytest

from conftest import golden_input
from oracle import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import sjpeg_amd as sj  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    assert sj.device_count() > 0, "no HIP device: the product path must fail loudly, not fall back"
    return sj.Engine(0)


def dev(img):
    return torch.from_numpy(np.ascontiguousarray(img)).cuda().unsqueeze(0)


def test_native_library_is_loaded():
    # the extension that runs is the in-tree HIP library, not a python/torch fallback
    maps = open("/proc/self/maps").read()
    sj.lib()
    maps = open("/proc/self/maps").read()
    assert "sjpeg_amd/csrc/libsjpeg_amd.so" in maps


def test_golden_small_vectors_host_api(golden_small):
    """SjpegEncode() (host pixels -> device -> bytes) vs the reference's bytes."""
    n = 0
    for key, want in golden_small.items():
        img, mode, q, method = golden_input(key)
        if method != 0:
            continue
        got = sj.SjpegEncode(img, q, 0, mode)
        assert got is not None, sj.last_error()
        assert got == want, key
        n += 1
    assert n >= 180


@pytest.mark.parametrize("mode", [1, 3, 4])
def test_coefficient_tap_matches_oracle(engine, oracle, mode):
    for (w, h, seed) in ((64, 64, 1), (250, 130, 2), (17, 13, 3), (1920, 1080, 4)):
        for gen in (synth.g_struct, synth.g_noise):
            img = gen(w, h, seed)
            for q in (35.0, 90.0):
                t, quant = sj.make_tables(quality=q)
                zz = engine.scan_coeffs(dev(img), t, mode)
                torch.cuda.synchronize()
                want = oracle.scan_coeffs(img, quant, 0x78, mode)
                assert (zz[0].cpu().numpy() == want).all(), (w, h, gen.__name__, q)


def test_random_sizes_and_qualities(engine, oracle):
    rng = np.random.RandomState(2024)
    for _ in range(80):
        w, h = int(rng.randint(1, 200)), int(rng.randint(1, 200))
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        if rng.rand() < 0.4:
            img[:] = img[:1, :1]                         # flat picture
        mode = int(rng.choice([1, 3, 4]))
        q = float(rng.choice([0, 3, 20, 50, 75, 95, 100]))
        got = sj.encode_device(dev(img), q, mode, engine=engine)[0]
        assert got == oracle.encode(img, q, mode), (w, h, mode, q)


def test_extreme_pixels(engine, oracle):
    # saturated checkerboards: largest coefficients, longest codes, many 0xFF bytes
    y, x = np.mgrid[0:96, 0:112]
    for period in (1, 2, 8):
        img = np.where((((x // period) + (y // period)) & 1)[..., None] == 1, 255, 0).astype(np.uint8)
        img = np.repeat(img, 3, axis=2)
        for mode in (1, 3, 4):
            for q in (100.0, 98.0, 50.0):
                got = sj.encode_device(dev(img), q, mode, engine=engine)[0]
                assert got == oracle.encode(img, q, mode), (period, mode, q)


def test_custom_matrices_min_quant_and_bias(engine, oracle):
    rng = np.random.RandomState(9)
    img = synth.g_struct(200, 120, 77)
    for _ in range(6):
        m = rng.randint(1, 256, (2, 64)).astype(np.uint8)
        mq = rng.randint(1, 40, (2, 64)).astype(np.uint8)
        bias = int(rng.randint(0, 256))
        t, quant = sj.make_tables(quant=m, min_quant=mq, q_bias=bias)
        header = sj.make_header(200, 120, 1, quant)
        out, sizes = engine.encode_frames(dev(img), t, header, 1)
        torch.cuda.synchronize()
        got = bytes(out[0, :int(sizes[0])].cpu().numpy())
        assert got == oracle.encode_matrices(img, m, min_quant=mq, q_bias=bias, yuv_mode=1)


def test_strides_padding_and_bottom_up(oracle):
    img = synth.g_struct(33, 21, 5)
    want = oracle.encode(img, 75.0, 1)
    padded = np.full((21, 33 * 3 + 29), 0xAB, np.uint8)           # padding bytes must not leak
    padded[:, :99] = img.reshape(21, 99)
    lib = sj.lib()
    out = C.POINTER(C.c_uint8)()
    n = lib.SjpegEncode(padded.ctypes.data, 33, 21, padded.strides[0], C.byref(out), 75.0, 0, 1)
    assert C.string_at(out, n) == want
    lib.SjpegFreeBuffer(out)
    flipped = img[::-1].copy()                                    # unit_test.cc:311-342
    got = sj.SjpegEncode(flipped, 75.0, 0, 1, stride=-flipped.strides[0])
    assert got == want


def test_batch_equals_individual_and_is_idempotent(engine, oracle):
    frames = np.stack([synth.g_struct(320, 176, 40 + k) for k in range(5)] +
                      [synth.g_struct(320, 176, 40)])
    got = sj.encode_device(torch.from_numpy(frames).cuda(), 75.0, 1, engine=engine)
    for k in range(6):
        assert got[k] == oracle.encode(frames[k], 75.0, 1)
    assert got[0] == got[5]
    again = sj.encode_device(torch.from_numpy(frames).cuda(), 75.0, 1, engine=engine)
    assert again == got


def test_output_slot_too_small_reports_zero(engine):
    img = synth.g_noise(256, 256, 1)
    t, quant = sj.make_tables(quality=95)
    header = sj.make_header(256, 256, 3, quant)
    out, sizes = engine.encode_frames(dev(img), t, header, 3, out_stride=4096)
    torch.cuda.synchronize()
    assert int(sizes[0]) == 0
    with pytest.raises(sj.SjpegError):
        engine.encode_frames(dev(img), t, header, 3, out_stride=16)


def test_abi_argument_errors(engine):
    img = dev(synth.g_struct(16, 16, 1))
    t, quant = sj.make_tables(quality=75)
    lib = sj.lib()
    sizes = torch.zeros(1, dtype=torch.int64, device="cuda")
    out = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
    args = lambda **kw: [kw.get("eng", engine._h), kw.get("rgb", img.data_ptr()), kw.get("rs", 48), 0,
                         kw.get("w", 16), kw.get("h", 16), kw.get("mode", 1), kw.get("n", 1),
                         C.byref(t), None, 0, 1, out.data_ptr(), 1 << 20, sizes.data_ptr(), None]
    assert lib.sjpeg_hip_encode_scan(*args()) == 0
    assert lib.sjpeg_hip_encode_scan(*args(rgb=None)) == -1
    assert lib.sjpeg_hip_encode_scan(*args(w=0)) == -1
    assert lib.sjpeg_hip_encode_scan(*args(w=65536)) == -1
    assert lib.sjpeg_hip_encode_scan(*args(mode=2)) == -1
    assert lib.sjpeg_hip_encode_scan(*args(rs=47)) == -1
    assert lib.sjpeg_hip_encode_scan(*args(n=0)) == -1
    assert lib.sjpeg_hip_encode_scan(*args(eng=None)) == -1           # NULL engine: an error code, not a crash
    assert b"" != lib.sjpeg_hip_last_error()
    torch.cuda.synchronize()


def test_unsupported_requests_fail_loudly():
    img = synth.g_struct(32, 32, 1)
    lib = sj.lib()
    if not lib.sjpeg_hip_has_riskiness_table() and not os.path.exists(os.path.join(sj.CSRC, "riskiness.bin")):
        # no table installed, none next to the library (tests/test_cxx_api.py isolates that case otherwise)
        assert sj.SjpegEncode(img, 75.0, 0, sj.YUV_AUTO) is None    # needs the reference's score table
        assert "not installed" in sj.last_error()
    out = C.POINTER(C.c_uint8)()
    assert lib.SjpegEncode(img.ctypes.data, 32, 32, 96, C.byref(out), 75.0, 0, 7) == 0   # bad mode


def test_concurrent_host_threads_are_deterministic(oracle):
    img = synth.g_struct(333, 211, 8)
    want = oracle.encode(img, 72.0, 1)
    res = [None] * 8

    def work(i):
        res[i] = sj.SjpegEncode(img, 72.0, 0, 1)

    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert all(r == want for r in res)


# ---- methods 1..6: adaptive quantization + optimised Huffman (default EncoderParam = 4) -------

@pytest.mark.parametrize("mode", [1, 3, 4])
def test_statistics_kernels_match_oracle(engine, oracle, mode):
    for (w, h, gen) in ((64, 64, synth.g_struct), (250, 130, synth.g_noise), (1920, 1080, synth.g_struct),
                        (17, 13, synth.g_noise)):
        img = gen(w, h, 31)
        hist = engine.scan_histogram(dev(img), mode)
        tables, quant = sj.make_tables(quality=80.0)
        freq = engine.scan_symbol_stats(dev(img), tables, mode)
        torch.cuda.synchronize()
        assert (hist[0].cpu().numpy().view(np.uint32) == oracle.histogram(img, mode)).all()
        want = oracle.symbol_stats(img, quant, 0x78, mode)
        assert (freq[0].cpu().numpy().view(np.uint32) == want).all()


def test_histogram_persistent_workgroups(oracle, monkeypatch):
    """The histogram kind is persistent (a workgroup bins several segments into 16-bit counters and leaves ONE
    partial): engines that may hold 1 / 3 / 7 / 50 workgroups at once walk many trips per workgroup, with group
    counts that do and do not divide the segments; a flat 4K picture puts every coefficient of 198 segments into
    one bin of a group's counters (the 16-bit limit is 266 segments)."""
    rng = np.random.RandomState(5)
    for mode, (w, h) in ((1, (1920, 1080)), (3, (1100, 700)), (4, (900, 1500))):
        imgs = [synth.g_struct(w, h, 41), rng.randint(0, 256, (h, w, 3)).astype(np.uint8), synth.g_struct(w, h, 43)]
        want = [oracle.histogram(im, mode) for im in imgs]
        frames = torch.from_numpy(np.stack(imgs)).cuda()
        for slots in (1, 3, 7, 50, 0):
            if slots:
                monkeypatch.setenv("SJPEG_HIP_HISTO_SLOTS", str(slots))
            else:
                monkeypatch.delenv("SJPEG_HIP_HISTO_SLOTS", raising=False)
            eng = sj.Engine(0)
            for rep in range(2):
                hist = eng.scan_histogram(frames, mode)
                torch.cuda.synchronize()
                for k in range(3):
                    assert (hist[k].cpu().numpy().view(np.uint32) == want[k]).all(), (mode, slots, k, rep)
            eng.close()
    # one large frame: more groups than one workgroup of the summing kernel walks -- its slices meet with atomics
    monkeypatch.delenv("SJPEG_HIP_HISTO_SLOTS", raising=False)
    big = synth.g_struct(7680, 4320, 47)
    eng = sj.Engine(0)
    hist = eng.scan_histogram(dev(big), 3)
    torch.cuda.synchronize()
    assert (hist[0].cpu().numpy().view(np.uint32) == oracle.histogram(big, 3)).all()
    eng.close()
    flat = np.full((2160, 3840, 3), 200, np.uint8)
    monkeypatch.setenv("SJPEG_HIP_HISTO_SLOTS", "1")
    eng = sj.Engine(0)
    hist = eng.scan_histogram(dev(flat), 1)
    torch.cuda.synchronize()
    assert (hist[0].cpu().numpy().view(np.uint32) == oracle.histogram(flat, 1)).all()
    eng.close()


def test_golden_methods_host_api(golden_small):
    n = 0
    for key, want in golden_small.items():
        img, mode, q, method = golden_input(key)
        if method in (1, 3, 4):
            got = sj.SjpegEncode(img, q, method, mode)
            assert got is not None, sj.last_error()
            assert got == want, key
            n += 1
    assert n == 9


def test_methods_random_vs_oracle(oracle):
    rng = np.random.RandomState(77)
    for _ in range(40):
        w, h = int(rng.randint(1, 160)), int(rng.randint(1, 160))
        img = synth.g_struct(w, h, int(rng.randint(1 << 30))) if rng.rand() < 0.6 else \
            rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        mode = int(rng.choice([1, 3, 4]))
        q = float(rng.choice([5, 40, 75, 92, 99]))
        m = int(rng.choice([1, 2, 3, 4, 5, 6]))
        got = sj.SjpegEncode(img, q, m, mode)
        assert got == oracle.encode_method(img, q, mode, m), (w, h, mode, q, m, sj.last_error())


def test_methods_full_size_digests(engine, digests):
    frames = dev(synth.g_struct(3840, 2160))
    for m in (1, 3, 4):
        got = sj.encode_device_method(frames, 75.0, 1, m, engine=engine)[0]
        d = digests[f"struct4k|420|q75|m{m}"]
        assert len(got) == d["size"] and hashlib.md5(got).hexdigest() == d["md5"], m
    host = sj.SjpegEncode(synth.g_struct(3840, 2160), 75.0, 4, 1)
    assert hashlib.md5(host).hexdigest() == digests["struct4k|420|q75|m4"]["md5"]


@pytest.mark.parametrize("key,gen,w,h,q,mode", [
    ("struct8k|444|q90", "struct", 7680, 4320, 90.0, 3),      # SURVEY 8c / BASELINE C3': cd6a30d5..., 24 565 823 B
    ("noise4k|420|q75", "noise", 3840, 2160, 75.0, 1),        # 1eb32ac3..., 4 445 849 B
    ("noise4k|444|q75", "noise", 3840, 2160, 75.0, 3),
    ("noise4k|400|q75", "noise", 3840, 2160, 75.0, 4),
    ("struct4k|444|q75", "struct", 3840, 2160, 75.0, 3),
    ("struct4k|400|q75", "struct", 3840, 2160, 75.0, 4),
])
def test_default_parameters_full_size_known_answers(engine, digests, key, gen, w, h, q, mode):
    """Default parameters (method 4: adaptive quantization + optimised Huffman tables,
    /root/reference/src/enc.cc:323-386, src/histogram.cc:126-339, src/entropy.cc:208-444) at full size in the
    modes and on the pictures where the statistics / replay kinds see wide levels (> 127), two bit windows per
    segment and 4:4:4 / 4:0:0 part lists: through the batch entry of the C-ABI (sjpeg_hip_encode_batch_src) and
    through the host API (what sjpeg::Encode does with a default EncoderParam and this colour mode)."""
    img = (synth.g_struct if gen == "struct" else synth.g_noise)(w, h)
    d = digests[key + "|m4"]
    got = sj.encode_device_method(dev(img), q, mode, 4, engine=engine)[0]
    assert len(got) == d["size"] and hashlib.md5(got).hexdigest() == d["md5"], key
    host = sj.SjpegEncode(img, q, 4, mode)
    assert host is not None, sj.last_error()
    assert len(host) == d["size"] and hashlib.md5(host).hexdigest() == d["md5"], key + " (host API)"
    if key.startswith("struct8k"):
        for m in (1, 3):                      # optimised tables alone, adaptive quantization alone
            dm = digests[f"{key}|m{m}"]
            got = sj.encode_device_method(dev(img), q, mode, m, engine=engine)[0]
            assert len(got) == dm["size"] and hashlib.md5(got).hexdigest() == dm["md5"], (key, m)


def test_default_parameters_batches_of_noise_and_444(engine, digests):
    """The same known answers as members of BATCHES (two-part batches: >= 125 Mpixels): four 8K 4:4:4 q90 frames
    (bench.py's "C3 default parameters x4"), and 4K noise next to the structured picture in one batch, so that a
    part holds frames with narrow and with wide kept blocks."""
    s8 = torch.from_numpy(synth.g_struct(7680, 4320)).cuda()
    frames = s8.unsqueeze(0).expand(4, -1, -1, -1).contiguous()
    got = sj.encode_device_method(frames, 90.0, 3, 4, engine=engine)
    d = digests["struct8k|444|q90|m4"]
    for k in range(4):
        assert len(got[k]) == d["size"] and hashlib.md5(got[k]).hexdigest() == d["md5"], k
    del frames, s8
    pics = [synth.g_noise(3840, 2160), synth.g_struct(3840, 2160)]
    frames = torch.from_numpy(np.stack([pics[k % 2] for k in range(18)])).cuda()
    for mode, mname in ((1, "420"), (3, "444")):
        got = sj.encode_device_method(frames, 75.0, mode, 4, engine=engine)
        for k in range(18):
            d = digests[("noise4k" if k % 2 == 0 else "struct4k") + f"|{mname}|q75|m4"]
            assert len(got[k]) == d["size"] and hashlib.md5(got[k]).hexdigest() == d["md5"], (mname, k)


def test_back_to_back_batches_without_a_host_wait(engine, oracle):
    """Two asynchronous batch calls of a method with optimised tables but no adaptive quantization (methods 1, 2: the
    call's first upload -- the statistics tables of its first part -- is not preceded by any host wait), two parts each
    (>= 24 frames), queued back to back: the second call's early uploads must not reach the table and header buffers
    before the first call's encode kernels have read them (scan_engine.hip: the upload stream waits for an
    event on the call's stream).  Both calls' frames equal the oracle's."""
    w, h = 1920, 1080
    pics = [synth.g_struct(w, h, 900 + k) for k in range(3)] + [synth.g_noise(w, h, 77)]
    want = {m: [oracle.encode_method(p, 75.0, 1, m) for p in pics] for m in (1, 2)}
    frames = torch.from_numpy(np.stack([pics[k % 4] for k in range(26)])).cuda()
    rows = frames.view(26, h, w * 3)
    src, _ = sj.make_source(sj.SRC_RGB, [rows])
    q = np.zeros((2, 64), np.uint8)
    sj.lib().sjpeg_hip_quality_matrices(75.0, q.ctypes.data)
    for piped in (False, True):
        engine.set_pipelined(piped)
        try:
            for rep in range(3):
                calls = [engine.encode_batch(src, 26, w, h, 1, q, method=m) for m in (1, 2, 1, 2)]
                engine.wait()
                torch.cuda.synchronize()
                for (out, sizes), m in zip(calls, (1, 2, 1, 2)):
                    got = sj._fetch_frames(out, sizes)
                    for k in range(26):
                        assert got[k] == want[m][k % 4], (piped, rep, m, k)
        finally:
            engine.set_pipelined(False)


def test_batch_lanes_many_jobs():
    """The lanes of the batch path (sjpeg_hip_encode_batch_src, round 6: jobs of a batch on up to four streams at once,
    child engines, one polling host thread) with small pictures: a process of its own with the job size set to 0.02
    Mpixels, so that a batch of 23 frames is many jobs, several per lane (tests/lanes_check.py); then once more with
    the lanes off (the two parts of round 5), the same answers."""
    import subprocess
    import sys
    for env_extra in ({"SJPEG_HIP_BATCH_JOB_MPIX": "0.02"}, {"SJPEG_HIP_BATCH_JOB_MPIX": "0.2", "SJPEG_HIP_BATCH_LANES": "3"},
                      {"SJPEG_HIP_BATCH_LANES": "0"}):
        env = dict(os.environ)
        env.update(env_extra)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lanes_check.py")], capture_output=True, text=True,
                           timeout=600, env=env)
        assert r.returncode == 0 and "lanes ok" in r.stdout, (env_extra, r.stdout[-500:], r.stderr[-2000:])
    # ... and the batch tests of this file themselves, every one with jobs of 0.01 Mpixels on the lanes: the other source
    # layouts (BGRA, RGBA, gray, planar, NV12 / NV21), per-frame tables, calls back to back, custom matrices with a floor
    if os.environ.get("SJPEG_LANES_INNER") is None:
        env = dict(os.environ)
        env.update({"SJPEG_HIP_BATCH_JOB_MPIX": "0.01", "SJPEG_LANES_INNER": "1"})
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q",
                            "-k", "batch_entry_other_layouts or methods_batched_per_frame_tables or c5_recompress_default or "
                                  "histogram_persistent or adaptive_decision"],
                           capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])


def test_methods_batched_per_frame_tables(engine, oracle):
    """One launch per pass over a batch whose frames each get their own adapted quantizer,
    optimised Huffman codes and header (sjpeg_hip_*_multi): every frame equals the reference's
    single-image encode."""
    rng = np.random.RandomState(4242)
    for (w, h, mode) in ((321, 203, 1), (160, 96, 3), (75, 131, 4), (1, 1, 1)):
        imgs = []
        for k in range(7):                      # very different content => different tables per frame
            if k % 3 == 0:
                imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))
            elif k % 3 == 1:
                imgs.append(synth.g_struct(w, h, 1000 + k))
            else:
                imgs.append(np.full((h, w, 3), 37 * k, np.uint8))
        frames = torch.from_numpy(np.stack(imgs)).cuda()
        for m, q in ((1, 60.0), (3, 85.0), (4, 75.0), (6, 30.0), (0, 75.0)):
            got = sj.encode_device_method(frames, q, mode, m, engine=engine)
            for k in range(7):
                assert got[k] == oracle.encode_method(imgs[k], q, mode, m), (w, h, mode, m, k)


def test_multi_rejects_bad_header_offsets(engine):
    import ctypes as C
    frames = dev(synth.g_struct(64, 48))
    rows = frames.view(1, 48, 64 * 3)
    src, _ = sj.make_source(sj.SRC_RGB, [rows])
    t, q = sj.make_tables(quality=75.0)
    hdr = sj.make_header(64, 48, 1, q)
    arr = (sj.ScanTables * 1)(t)
    out = torch.empty((1, 65536), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(1, dtype=torch.int64, device="cuda")
    offs = (C.c_size_t * 2)(len(hdr), 0)        # descending
    rc = sj.lib().sjpeg_hip_encode_scan_multi(engine._h, C.byref(src), 64, 48, 1, 1, C.cast(arr, C.c_void_p), hdr,
                                             offs, 1, out.data_ptr(), 65536, sizes.data_ptr(), None)
    assert rc != 0 and "header_offsets" in sj.lib().sjpeg_hip_last_error().decode()


def test_row_offsets_beyond_2_31_bytes(engine, oracle):
    """64-bit addressing: rows whose byte offset in the source passes 2^31 (and 2^32).  The same
    pixels coded from a contiguous buffer are the expected bytes (a stride never changes the
    output).  The reference itself addresses MCUs with 32-bit ints (src/encoders.cc:171,207,240)
    and is not a checker past 2^31: profiles/HISTORY_r01.md, tools/huge_frame_vs_oracle.py."""
    w, h = 200, 120
    img = synth.g_struct(w, h, 2468)
    stride = 40 * 1000 * 1000 + 16                 # row 54 starts beyond 2^31, row 108 beyond 2^32
    big = torch.empty(h * stride, dtype=torch.uint8, device="cuda")
    rows = big.as_strided((1, h, w * 3), (h * stride, stride, 1))
    rows.copy_(torch.from_numpy(img).cuda().view(1, h, w * 3))
    t, q = sj.make_tables(quality=80.0)
    for mode in (1, 3, 4):
        src, _ = sj.make_source(sj.SRC_RGB, [rows])
        out, sizes = engine.encode_source(src, 1, w, h, t, sj.make_header(w, h, mode, q), mode)
        torch.cuda.synchronize()
        got = bytes(out[0, :int(sizes[0])].cpu().numpy())
        assert got == oracle.encode(img, 80.0, mode), mode
    # a planar source with the same property (luma plane of a 4:2:0 source)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    rng = np.random.RandomState(3)
    yp = rng.randint(0, 256, (h, w)).astype(np.uint8)
    up = rng.randint(0, 256, (ch, cw)).astype(np.uint8)
    vp = rng.randint(0, 256, (ch, cw)).astype(np.uint8)
    ybig = big.as_strided((1, h, w), (h * stride, stride, 1))
    ybig.copy_(torch.from_numpy(yp).cuda().view(1, h, w))
    want = sj.encode_source_method(5, [torch.from_numpy(yp).cuda().unsqueeze(0), torch.from_numpy(up).cuda().unsqueeze(0),
                                       torch.from_numpy(vp).cuda().unsqueeze(0)], w, h, 80.0, 1, 0, engine=engine)
    got = sj.encode_source_method(5, [ybig, torch.from_numpy(up).cuda().unsqueeze(0),
                                      torch.from_numpy(vp).cuda().unsqueeze(0)], w, h, 80.0, 1, 0, engine=engine)
    assert got == want
    del big


def test_pipelined_calls_same_bytes(oracle):
    """Back-to-back encode calls in pipelined mode (stitch of call i under K1 of call i + 1, two
    sets of segment buffers): every call's output equals the oracle, whatever follows it --
    different pictures, geometries and batch sizes in consecutive calls, a statistics call in
    between, and switching the mode off again."""
    eng = sj.Engine(0)
    eng.set_pipelined(True)
    rng = np.random.RandomState(99)
    jobs, keep = [], []
    for it in range(9):
        w, h = [(640, 360), (333, 217), (1280, 720), (64, 48)][it % 4]
        f = 1 + it % 3
        mode = (1, 3, 4)[it % 3]
        imgs = [synth.g_struct(w, h, 500 + 10 * it + k) if (it + k) % 2 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
                for k in range(f)]
        frames = torch.from_numpy(np.stack(imgs)).cuda()
        t, q = sj.make_tables(quality=70.0 + it)
        hdr = sj.make_header(w, h, mode, q)
        out, sizes = eng.encode_frames(frames, t, hdr, mode)          # own output buffers per call
        keep.append(frames)
        jobs.append((imgs, 70.0 + it, mode, out, sizes))
        if it == 4:                                                  # an ordered entry point in the middle
            eng.scan_symbol_stats(frames, t, mode)
    eng.wait()
    torch.cuda.synchronize()
    for (imgs, q, mode, out, sizes) in jobs:
        sz = sizes.cpu().numpy()
        for k, img in enumerate(imgs):
            assert bytes(out[k, :int(sz[k])].cpu().numpy()) == oracle.encode(img, q, mode), (img.shape, q, mode, k)
    eng.set_pipelined(False)
    img = synth.g_struct(321, 123, 5)
    assert sj.encode_device(torch.from_numpy(img).cuda().unsqueeze(0), 75.0, 1, engine=eng)[0] == oracle.encode(img, 75.0, 1)


def test_non_default_streams(oracle):
    """Everything is ordered on the stream the caller passes, and the engine orders calls that
    arrive on different streams (they share its scratch): two torch streams alternate on one
    engine, inputs are produced on the stream that codes them."""
    eng = sj.Engine(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    img = [synth.g_struct(500, 300, 40 + k) for k in range(6)]
    host = [torch.from_numpy(im).pin_memory() for im in img]
    t, q = sj.make_tables(quality=82.0)
    hdr = sj.make_header(500, 300, 1, q)
    outs = []
    for k in range(6):
        st = s1 if k % 2 == 0 else s2
        with torch.cuda.stream(st):
            frames = host[k].to("cuda", non_blocking=True).unsqueeze(0)      # the upload is on the same stream
            out, sizes = eng.encode_frames(frames, t, hdr, 1)
            outs.append((frames, out, sizes))
    s1.synchronize()
    s2.synchronize()
    for k, (_, out, sizes) in enumerate(outs):
        assert bytes(out[0, :int(sizes[0])].cpu().numpy()) == oracle.encode(img[k], 82.0, 1), k


def test_c5_recompress_default_params(engine, digests):
    d = digests["recompress|r90|default"]
    src = np.array(digests["recompress|r90|m0"]["source_quant"], np.uint8).reshape(2, 64)
    quant = np.clip((src.astype(np.float64) * 100.0 / 90.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)
    got = sj.encode_device_method(dev(synth.g_struct(3840, 2160)), yuv_mode=1, method=4, engine=engine,
                                  quant=quant, min_quant=quant)[0]
    assert len(got) == d["size"] and hashlib.md5(got).hexdigest() == d["md5"]


def test_c5_end_to_end_4k(engine, digests):
    """BASELINE config C5 as the reference's CLI runs it (examples/sjpeg.cc:262-286), every step by the
    product: encode the 4K source at q 92 with default parameters, read ITS quantizer back with
    SjpegFindQuantizer, recompress the pixels at reduction 90 (SetQuantization + SetLimitQuantization)
    with default parameters and with method 0 -- three known answers of SURVEY 8c."""
    import ctypes as C
    frames = dev(synth.g_struct(3840, 2160))
    source = sj.encode_device_method(frames, quality=92.0, yuv_mode=1, method=4, engine=engine)[0]
    d = digests["recompress|source_q92_default"]
    assert len(source) == d["size"] and hashlib.md5(source).hexdigest() == d["md5"]
    found = np.zeros((2, 64), np.uint8)
    assert sj.lib().SjpegFindQuantizer(source, len(source), found.ctypes.data) == d["nq"]
    assert found.reshape(-1).tolist() == digests["recompress|r90|m0"]["source_quant"]
    quant = np.clip((found.astype(np.float64) * 100.0 / 90.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)
    for key, method in (("recompress|r90|default", 4), ("recompress|r90|m0", 0)):
        got = sj.encode_device_method(frames, yuv_mode=1, method=method, engine=engine, quant=quant, min_quant=quant)[0]
        assert len(got) == digests[key]["size"] and hashlib.md5(got).hexdigest() == digests[key]["md5"], key


def test_batch_in_parts_equals_per_picture_encodes(engine, oracle):
    """sjpeg_hip_encode_batch_src codes a batch of 24 frames or more in two parts (host analysis of one under
    the device pass of the other, partial sums on a side stream, kept blocks addressed by part): every
    frame must still be the reference's single-picture encode -- odd frame counts, every analysis method."""
    rng = np.random.RandomState(77)
    for (w, h, f) in ((160, 96, 27), (97, 61, 24), (320, 200, 33)):
        imgs = [synth.g_struct(w, h, 3000 + k) if k % 3 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8) for k in range(f)]
        frames = torch.from_numpy(np.stack(imgs)).cuda()
        for mode, q, method in ((1, 75.0, 4), (3, 90.0, 1), (4, 50.0, 3), (1, 30.0, 6)):
            got = sj.encode_device_method(frames, q, mode, method, engine=engine)
            for k in range(f):
                assert got[k] == oracle.encode_method(imgs[k], q, mode, method), (w, h, f, k, mode, method)
    # twice in a row on one engine, then a smaller (single-part) batch: the part state does not leak
    got = sj.encode_device_method(frames[:5], 75.0, 1, 4, engine=engine)
    for k in range(5):
        assert got[k] == oracle.encode_method(imgs[k], 75.0, 1, 4), k


def test_kept_blocks_narrow_and_wide(engine, oracle):
    """The statistics pass keeps a quantized block as bytes when no AC level exceeds 127 and adds a second plane
    of high bytes when one does (scan_segments.h); the replay kind rebuilds the 16-bit entries.  A batch that mixes
    both kinds of block in one workgroup -- noise at q 97..100 (levels of several hundred) beside flat and smooth
    pictures, DC differences beyond 127 beside tiny ones, negative coefficients that quantize to zero -- must still
    be the reference's per-picture encodes, with the statistics pass fed from the pixels (methods 1, 2: no
    histogram pass) and from the histogram pass' coefficients (methods 4, 6)."""
    rng = np.random.RandomState(4242)
    for (w, h) in ((64, 48), (97, 61), (336, 200)):
        imgs = []
        for k in range(6):
            if k % 3 == 0:
                imgs.append(rng.randint(0, 256, (h, w, 3)).astype(np.uint8))                     # noise: wide blocks at high q
            elif k % 3 == 1:
                img = synth.g_struct(w, h, 900 + k)
                img[: h // 2, : w // 2] = rng.randint(0, 256, (h // 2, w // 2, 3))              # both kinds in one picture
                imgs.append(img)
            else:
                img = np.zeros((h, w, 3), np.uint8)
                img[:, ::16] = 255                                                                # large DC steps, sparse AC
                img[::7] //= 2
                imgs.append(img)
        frames = torch.from_numpy(np.stack(imgs)).cuda()
        for mode, q, method in ((1, 100.0, 4), (3, 98.0, 6), (4, 100.0, 2), (1, 97.0, 1), (3, 100.0, 4), (1, 60.0, 4)):
            got = sj.encode_device_method(frames, q, mode, method, engine=engine)
            for k in range(len(imgs)):
                assert got[k] == oracle.encode_method(imgs[k], q, mode, method), (w, h, k, mode, q, method)


def test_batches_over_the_scratch_limit_go_in_several_launches(oracle, monkeypatch):
    """A batch whose segment scratch would pass the engine's limit (SJPEG_HIP_SCRATCH_LIMIT_BYTES, read when the engine
    is made; 16 GiB by default) is coded as several launches of as many frames as fit: same bytes, one frame per launch
    (limit 1) or a few, ordered and pipelined, one header for all and a header per frame (default parameters)."""
    imgs = [synth.g_struct(176, 96, 300 + k) for k in range(7)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    want = [oracle.encode(im, 80.0, 1) for im in imgs]
    want4 = [oracle.encode_method(im, 80.0, 1, 4) for im in imgs]
    for limit in ("1", "150000", "400000"):
        monkeypatch.setenv("SJPEG_HIP_SCRATCH_LIMIT_BYTES", limit)
        eng = sj.Engine(0)
        for piped in (False, True):
            eng.set_pipelined(piped)
            assert sj.encode_device(frames, 80.0, 1, engine=eng) == want, (limit, piped)
            assert sj.encode_device(frames, 80.0, 1, engine=eng) == want, (limit, piped)
        eng.set_pipelined(False)
        assert sj.encode_device_method(frames, 80.0, 1, 4, engine=eng) == want4, limit


def _pictures_of_sparse_blocks(rng, bw, bh, q):
    """Gray pictures whose 8x8 blocks are inverse DCTs of chosen sparse coefficient patterns (levels +-1..3 times
    the quantizer step q: they survive the rounding to 8 bits, everything else stays zero), built to sit ON the rules
    by which K1 codes two quarters of the zig-zag scan as one part (scan_segments.h, `mg01` / `mg23`): a run of exactly
    15 / 16 zeros across the quarter boundary, 16 / 17 symbols in the two quarters, an empty first quarter, position 63."""
    from scipy.fft import idctn
    zig = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
           28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    img = np.zeros((bh * 8, bw * 8), np.uint8)
    for by in range(bh):
        for bx in range(bw):
            pos = set()
            kind = rng.randint(0, 9)
            half = 32 * rng.randint(0, 2)                         # the rule applies to quarters 0 + 1 and to 2 + 3 alike
            if kind == 0:                                         # run of 15 / 16 across the quarter boundary
                last = rng.randint(1, 16)
                pos = {half + last, half + last + 16 + rng.randint(0, 2)} - {half + 0}
            elif kind == 1:                                       # 16 / 17 symbols in the two quarters
                n = 16 + rng.randint(0, 2)
                pos = set(half + p for p in rng.choice(np.arange(1, 32), n, replace=False))
                pos |= {half + 15, half + 16}                     # (no long run between the quarters)
            elif kind == 2:                                       # first quarter empty: run 15 / 16 from the DC (or from quarter 1)
                pos = {half + 16 + rng.randint(0, 2)} | set(half + p for p in rng.choice(np.arange(18, 32), rng.randint(0, 5), replace=False))
            elif kind == 3:                                       # both halves merged, and position 63 (no EOB)
                pos = {1, 14, 17, 30, 33, 47, 49, 63} if rng.randint(0, 2) else {3, 15, 16, 40, 47, 48}
            elif kind == 4:                                       # dense block: nothing merges
                pos = set(rng.choice(np.arange(1, 64), rng.randint(34, 63), replace=False))
            elif kind == 5:                                       # DC only / empty
                pos = set()
            else:                                                 # ordinary sparse blocks
                pos = set(rng.choice(np.arange(1, 64), rng.randint(1, 14), replace=False) % (rng.choice([16, 32, 48, 64])))
            pos = {p for p in pos if 1 <= p <= 63}
            F = np.zeros(64)
            for p in pos:
                F[zig[p]] = q * rng.choice([-3, -2, -1, 1, 2, 3], p=[0.05, 0.1, 0.35, 0.35, 0.1, 0.05])
            F[0] = q * rng.randint(-6, 7)
            blk = idctn(F.reshape(8, 8), norm="ortho") + 128.0
            img[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = np.clip(np.rint(blk), 0, 255)
    return img


def test_two_quarters_as_one_part(engine, oracle):
    """K1 codes two quarters of a block's zig-zag scan as ONE part when the lean walk can take them in one go: both hold
    symbols, at most 16 between them, and fewer than 16 zeros between the last symbol of the first and the first of the
    second (so that only the part's first symbol can need ZRL codes).  Pictures made of blocks that sit on those three
    rules -- checked on the oracle's own coefficients, so that the test cannot go blind --, as gray, and as R = G = B
    through the colour kinds (chroma blocks: a DC and nothing else), one frame and a batch, standard and optimised
    tables (replay kind)."""
    rng = np.random.RandomState(20260929)
    q = 16
    quant = np.full((2, 64), q, np.uint8)
    pics = [_pictures_of_sparse_blocks(rng, 41, 13, q) for _ in range(3)] + [_pictures_of_sparse_blocks(rng, 12, 7, q)]
    # coverage, on the reference's coefficients: every rule is met from both sides in these pictures
    seen = {"run15": 0, "run16": 0, "sum16": 0, "sum17": 0, "empty_first": 0, "pos63": 0, "both_halves": 0}
    for img in pics:
        h, w = img.shape
        zz = oracle.scan_coeffs(np.repeat(img[..., None], 3, 2), quant, yuv_mode=4)
        nz = zz != 0
        nz[:, 0] = False
        for b in nz:
            merged = 0
            for half in (0, 32):
                a, c = b[half:half + 16], b[half + 16:half + 32]
                if not c.any():
                    continue
                end_a = (np.nonzero(a)[0].max() + 1) if a.any() else (1 if half == 0 else 0)
                run = 16 + np.nonzero(c)[0].min() - end_a if (a.any() or half == 0) else None
                if run == 15: seen["run15"] += 1
                if run == 16: seen["run16"] += 1
                if run is not None and run < 16:
                    n = int(a.sum() + c.sum())
                    if n == 16: seen["sum16"] += 1
                    if n == 17: seen["sum17"] += 1
                    if not a.any() and half == 0: seen["empty_first"] += 1
                    if n <= 16: merged += 1
            if b[63]: seen["pos63"] += 1
            if merged == 2: seen["both_halves"] += 1
    assert all(v >= 3 for v in seen.values()), seen
    for img in pics:
        h, w = img.shape
        gray = torch.from_numpy(img).cuda().unsqueeze(0)
        want = oracle.encode_src(sj.SRC_GRAY, [img], w, h, quant, yuv_mode=4)
        assert sj.encode_source_method(sj.SRC_GRAY, [gray], w, h, 75.0, 4, 0, engine=engine, quant=quant) == want
        for method in (1, 4):                                 # optimised tables: statistics + replay kinds
            assert (sj.encode_source_method(sj.SRC_GRAY, [gray], w, h, 75.0, 4, method, engine=engine, quant=quant) ==
                    oracle.encode_src(sj.SRC_GRAY, [img], w, h, quant, yuv_mode=4, method=method)), method
        rgb = np.repeat(img[..., None], 3, 2).copy()
        for mode in (1, 3):
            got = sj.encode_device(torch.from_numpy(rgb).cuda().unsqueeze(0), 75.0, mode, engine=engine, quant=quant)[0]
            assert got == oracle.encode_matrices(rgb, quant, yuv_mode=mode), mode
    # a batch (one launch, parts of several frames in flight), default parameters
    frames = torch.from_numpy(np.stack([np.repeat(p[..., None], 3, 2) for p in pics[:3]])).cuda()
    got = sj.encode_device_method(frames, 75.0, 1, 4, engine=engine, quant=quant)
    for k in range(3):
        assert got[k] == oracle.encode_full(np.repeat(pics[k][..., None], 3, 2), quant, yuv_mode=1, method=4), k


def test_saturated_primaries_chroma_plus_128(engine, oracle):
    """Pure blue makes Cb = +128 and pure red Cr = +128 (one past the int8 range the other samples stay in):
    four such columns in a block put the row pass' even sum at 32768, one past int16 -- solid red and blue
    pictures were wrong in 4:2:0 and 4:4:4 until this was accumulated in 32 bits (found by the fuzz, round 3)."""
    rng = np.random.RandomState(12)
    prim = np.array([(0, 0, 255), (255, 0, 0), (255, 0, 255), (0, 255, 0), (255, 255, 255), (0, 0, 0), (0, 255, 255), (255, 255, 0)], np.uint8)
    pics = []
    for px in prim[:3]:
        for (w, h) in ((8, 8), (16, 16), (64, 8), (100, 37)):
            img = np.zeros((h, w, 3), np.uint8)
            img[:] = px
            pics.append(img)
    for (w, h, cell) in ((257, 1, 1), (640, 1, 1), (96, 64, 8), (160, 120, 16), (333, 77, 4)):   # primaries in cells / columns
        idx = rng.randint(0, 8, ((h + cell - 1) // cell, (w + cell - 1) // cell))
        pics.append(np.repeat(np.repeat(prim[idx], cell, 0), cell, 1)[:h, :w].copy())
    for img in pics:
        h, w = img.shape[:2]
        for mode in (1, 3, 4):
            for q, method in ((75.0, 0), (99.0, 0), (90.0, 4), (100.0, 1), (50.0, 3), (99.0, 7)):
                assert sj.SjpegEncode(img, q, method, mode) == oracle.encode_method(img, q, mode, method), (w, h, mode, q, method)
        for fmt in (1, 2):                                   # BGRA / RGBA through the 4-byte P1
            x = np.zeros((h, w, 4), np.uint8)
            x[..., :3] = img[..., ::-1] if fmt == 1 else img
            planes = [x.reshape(h, 4 * w)]
            got = sj.encode_source_method(fmt, [torch.from_numpy(p).cuda().unsqueeze(0) for p in planes], w, h, 99.0, 3, 4, engine=engine)
            assert got == oracle.encode_src(fmt, planes, w, h, oracle.quality_matrices(99.0), yuv_mode=3, method=4), (fmt, w, h)


def test_flat_pictures_with_one_bit_codes(oracle):
    """Flat pictures coded with optimised tables are streams of one-bit codes: segments of a few dozen
    bits, a last segment that may start no 32-bit word of its own -- then the frame ends inside an
    EARLIER segment's last word, whose padding bytes must not be counted as 0xFF data (found by the
    soak: sizes 1-3 bytes too long, zeros in front of the EOI)."""
    for (w, h, q, method, mode) in ((123, 251, 75.0, 2, 4), (203, 158, 75.0, 6, 4), (193, 296, 0.0, 4, 1),
                                    (219, 1356, 100.0, 2, 1), (131, 34, 100.0, 7, 3), (2027, 7, 3.0, 2, 3),
                                    (2044, 30, 97.0, 2, 4), (1161, 2095, 90.0, 3, 4), (60, 250, 0.0, 3, 4),
                                    (16, 16, 75.0, 1, 1), (8, 8, 50.0, 4, 4), (4000, 24, 75.0, 1, 1)):
        for v in (0, 3, 77, 128, 200, 255):
            img = np.full((h, w, 3), v, np.uint8)
            got = sj.SjpegEncode(img, q, method, mode)
            assert got == oracle.encode_method(img, q, mode, method), (w, h, q, method, mode, v)
            if method <= 6:
                got = sj.encode_device_method(dev(img), q, mode, method)[0]
                assert got == oracle.encode_method(img, q, mode, method), ("batch", w, h, q, method, mode, v)


# ---- other input layouts: BGRA / RGBA / gray / planar YUV / NV12 / NV21 -------------------------

def _random_planes(rng, fmt, w, h):
    cw, ch = (w + 1) // 2, (h + 1) // 2
    shapes = {1: [(h, 4 * w)], 2: [(h, 4 * w)], 3: [(h, w)], 4: [(h, w)] * 3, 5: [(h, w), (ch, cw), (ch, cw)],
              6: [(h, w), (ch, 2 * cw)], 7: [(h, w), (ch, 2 * cw)]}[fmt]
    return [rng.randint(0, 256, s).astype(np.uint8) for s in shapes]


@pytest.mark.parametrize("fmt", [1, 2, 3, 4, 5, 6, 7])
def test_source_layouts_vs_oracle(engine, oracle, fmt):
    rng = np.random.RandomState(100 + fmt)
    for (w, h) in ((1, 1), (16, 16), (17, 13), (40, 9), (97, 61), (250, 130), (1920, 1080)):
        planes = _random_planes(rng, fmt, w, h)
        if w >= 97:                                   # smoother content for the big ones
            planes = [(p // 4 + np.arange(p.shape[1])[None, :] // 3).astype(np.uint8) for p in planes]
        dev_planes = [torch.from_numpy(p).cuda().unsqueeze(0) for p in planes]
        modes = (1, 3, 4) if fmt in (1, 2) else (1,)
        for mode in modes:
            for q, method in ((75.0, 0), (40.0, 4), (92.0, 3), (60.0, 1)):
                got = sj.encode_source_method(fmt, dev_planes, w, h, q, mode, method, engine=engine)
                want = oracle.encode_src(fmt, planes, w, h, oracle.quality_matrices(q), yuv_mode=mode,
                                         method=method)
                assert got == want, (fmt, w, h, mode, q, method)


@pytest.mark.parametrize("fmt", [1, 3, 4, 5, 6, 7])
def test_batch_entry_other_layouts(engine, oracle, fmt):
    """sjpeg_hip_encode_batch_src over batches in the other source layouts: every frame equals the
    oracle's single-picture encode with the same method."""
    rng = np.random.RandomState(700 + fmt)
    implied = {3: 4, 4: 3, 5: 1, 6: 1, 7: 1}
    for (w, h, f) in ((33, 21, 3), (250, 130, 4)):
        per_frame = [_random_planes(rng, fmt, w, h) for _ in range(f)]
        if w >= 97:
            per_frame = [[(p // 4 + np.arange(p.shape[1])[None, :] // 3 + 7 * k).astype(np.uint8) for p in planes]
                         for k, planes in enumerate(per_frame)]
        stacked = [torch.from_numpy(np.stack([pf[i] for pf in per_frame])).cuda() for i in range(len(per_frame[0]))]
        src, n = sj.make_source(fmt, stacked)
        assert n == f
        mode = implied.get(fmt, 1)
        for q, method in ((75.0, 4), (50.0, 1), (90.0, 3), (75.0, 0)):
            out, sizes = engine.encode_batch(src, f, w, h, mode, oracle.quality_matrices(q), method)
            torch.cuda.synchronize()
            sz = sizes.cpu().numpy()
            for k in range(f):
                want = oracle.encode_src(fmt, per_frame[k], w, h, oracle.quality_matrices(q), yuv_mode=mode, method=method)
                assert bytes(out[k, :int(sz[k])].cpu().numpy()) == want, (fmt, w, h, k, q, method)
    with pytest.raises(sj.SjpegError):
        engine.encode_batch(src, f, w, h, mode, oracle.quality_matrices(75.0), 7)


@pytest.mark.parametrize("fmt", [0, 1, 3, 4, 5, 6])
def test_search_pass_measurements_vs_oracle(engine, oracle, fmt):
    """What one pass of the size / PSNR search measures on the device: the squared quantization
    error (reference src/dichotomy.cc:309-323) and the BitCounter's bit count with the standard
    tables (src/bit_writer.h:292-365) rebuilt from the coded size + the entropy-bit total."""
    rng = np.random.RandomState(300 + fmt)
    for (w, h) in ((1, 1), (17, 13), (97, 61), (640, 360)):
        if fmt == 0:
            planes = [rng.randint(0, 256, (h, 3 * w)).astype(np.uint8)]
        else:
            planes = _random_planes(rng, fmt, w, h)
        if w >= 97:
            planes = [(p // 4 + np.arange(p.shape[1])[None, :] // 3).astype(np.uint8) for p in planes]
        dev_planes = [torch.from_numpy(p).cuda().unsqueeze(0) for p in planes]
        src, n = sj.make_source(fmt, dev_planes)
        modes = (1, 3, 4) if fmt in (0, 1) else ({3: 4, 4: 3}.get(fmt, 1),)
        for mode in modes:
            for q in (8.0, 60.0, 97.0):
                tables, quant = sj.make_tables(quality=q)
                err = engine.scan_quant_error_source(src, n, w, h, tables, mode)
                assert int(err[0].item()) == oracle.quant_error(fmt, planes, w, h, quant, yuv_mode=mode), (fmt, w, h, mode, q)
                out, sizes = engine.encode_source(src, n, w, h, tables, b"", mode)
                bits = int(engine.entropy_bits(1)[0])
                coded = bytes(out[0, :int(sizes[0].item())].cpu().numpy())
                body = coded[:-2]                                        # drop EOI
                escapes = len(body) - (bits + 7) // 8
                if bits % 8 != 0 and body[-2:] == b"\xff\x00":
                    escapes -= 1
                assert bits + 8 * escapes == oracle.counted_bits(fmt, planes, w, h, quant, yuv_mode=mode), (fmt, w, h, mode, q)


def test_source_argument_errors(engine):
    y = torch.zeros((1, 16, 16), dtype=torch.uint8, device="cuda")
    t, quant = sj.make_tables(quality=75)
    src, _ = sj.make_source(sj.SRC_GRAY, [y])
    with pytest.raises(sj.SjpegError):                 # gray implies 4:0:0
        engine.encode_source(src, 1, 16, 16, t, b"", sj.YUV_420)
    src, _ = sj.make_source(sj.SRC_NV12, [y])           # chroma plane missing
    with pytest.raises(sj.SjpegError):
        engine.encode_source(src, 1, 16, 16, t, b"", sj.YUV_420)
    c = torch.zeros((1, 8, 4), dtype=torch.uint8, device="cuda")
    src, _ = sj.make_source(sj.SRC_YUV420, [y, c, c])   # chroma rows shorter than (16 + 1) / 2
    with pytest.raises(sj.SjpegError):
        engine.encode_source(src, 1, 16, 16, t, b"", sj.YUV_420)
    torch.cuda.synchronize()


# ---- BASELINE.json full-size configurations --------------------------------------------------

@pytest.mark.parametrize("name,mname", [("struct4k", "420"), ("noise4k", "420"), ("struct4k", "444"),
                                        ("noise4k", "444"), ("struct4k", "400"), ("noise4k", "400")])
def test_c2_4k_digests(engine, digests, name, mname):
    gen = synth.g_struct if name.startswith("struct") else synth.g_noise
    mode = {"420": 1, "444": 3, "400": 4}[mname]
    got = sj.encode_device(dev(gen(3840, 2160)), 75.0, mode, engine=engine)[0]
    d = digests[f"{name}|{mname}|q75|m0"]
    assert len(got) == d["size"] and hashlib.md5(got).hexdigest() == d["md5"]
    # size-independent properties: a decoder accepts it and sees the right geometry
    from PIL import Image
    im = Image.open(io.BytesIO(got))
    assert im.size == (3840, 2160)
    im.load()
    assert got[:2] == b"\xff\xd8" and got[-2:] == b"\xff\xd9"
    body = got[got.index(b"\xff\xda") + 14 if mname != "400" else got.index(b"\xff\xda") + 10:-2]
    ff = [i for i in range(len(body) - 1) if body[i] == 0xFF]
    assert all(body[i + 1] == 0 for i in ff[:100000])              # every 0xFF is stuffed


def test_c3_8k_444_q90(engine, digests):
    got = sj.encode_device(dev(synth.g_struct(7680, 4320)), 90.0, 3, engine=engine)[0]
    d = digests["struct8k|444|q90|m0"]
    assert len(got) == d["size"] and hashlib.md5(got).hexdigest() == d["md5"]


@pytest.mark.parametrize("key,gen,w,h,q,mode,stride_mb", [
    ("noise4k|444|q75|m0", "noise", 3840, 2160, 75.0, 3, 16),     # 10.3 MB of stream: 2 503 chunks, 791 x 3 / ... segments
    ("noise4k|420|q75|m0", "noise", 3840, 2160, 75.0, 1, 12),     # 4.9 MB in a 12 MB slot: 3 072 chunks of room
    ("struct8k|444|q90|m0", "struct", 7680, 4320, 90.0, 3, 64),   # 25.5 MB, 6 172 segments, 6 204 chunks
    ("struct8k|444|q90|m0", "struct", 7680, 4320, 90.0, 3, 26),   # the same frame with 2 % of room behind it
])
def test_one_large_frame_sums_on_demand(engine, digests, key, gen, w, h, q, mode, stride_mb):
    """ONE frame whose stream budget is beyond the scans-in-LDS of the small fused stitch (more than 2 048 chunks or
    segments): K2 / K4 run inside K3 / K5 in their on-demand form (place_segments<2>, stuff_chunks<2>: every workgroup
    adds up the lengths / 0xFF counts in front of its own, round 6) -- the reference's bytes for slots of several sizes
    (/root/reference/src/bit_writer.h:172-209: flush granularity does not change the output)."""
    img = (synth.g_struct if gen == "struct" else synth.g_noise)(w, h)
    tables, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, mode, quant)
    out, sizes = engine.encode_frames(dev(img), tables, header, mode, out_stride=stride_mb << 20)
    torch.cuda.synchronize()
    n = int(sizes[0].item())
    d = digests[key]
    assert n == d["size"] and hashlib.md5(bytes(out[0, :n].cpu().numpy())).hexdigest() == d["md5"], (key, stride_mb)
    # a slot the stream does not fit: size 0 by contract, from the same kernels
    out2, sizes2 = engine.encode_frames(dev(img), tables, header, mode, out_stride=((d["size"] * 3 // 4) + 4095) & ~4095)
    torch.cuda.synchronize()
    assert int(sizes2[0].item()) == 0


def test_c4_batch_64x1080p(engine, digests):
    frames = torch.empty((64, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
    for k in range(64):
        frames[k] = torch.from_numpy(synth.g_struct(1920, 1080, 7654321 + k)).cuda()
    t, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(1920, 1080, 1, quant)
    out, sizes = engine.encode_frames(frames, t, header, 1, out_stride=4 << 20)
    torch.cuda.synchronize()
    sz = sizes.cpu().numpy()
    cat = hashlib.md5()
    for k in range(64):
        b = bytes(out[k, :int(sz[k])].cpu().numpy())
        cat.update(b)
        if k in (0, 1, 63):
            d = digests[f"struct1080p_k{k}|420|q75|m0"]
            assert len(b) == d["size"] and hashlib.md5(b).hexdigest() == d["md5"]
    d = digests["struct1080p_k0..63_concat|420|q75|m0"]
    assert int(sz.sum()) == d["size"] and cat.hexdigest() == d["md5"]


def test_c5_recompress_method0(engine, digests):
    d = digests["recompress|r90|m0"]
    src = np.array(d["source_quant"], np.uint8).reshape(2, 64)
    quant = np.clip((src.astype(np.float64) * 100.0 / 90.0 + 0.5).astype(np.int64), 1, 255).astype(np.uint8)
    t, fq = sj.make_tables(quant=quant, min_quant=quant)
    header = sj.make_header(3840, 2160, 1, fq)
    out, sizes = engine.encode_frames(dev(synth.g_struct(3840, 2160)), t, header, 1)
    torch.cuda.synchronize()
    got = bytes(out[0, :int(sizes[0])].cpu().numpy())
    assert len(got) == d["size"] and hashlib.md5(got).hexdigest() == d["md5"]


# ---- one frame over several devices: bands of segments, simulated on one GPU ----

def test_stitch_bands_arbitrary_cuts(engine, oracle):
    """The root's half of the banded path: bit strings cut at arbitrary bit positions (empty and
    few-bit bands included) come back as the reference's entropy segment, byte for byte."""
    from test_dist_cpu import _pack_words, _unstuffed_bits
    rng = np.random.RandomState(5)
    for (w, h, q, mode) in ((200, 120, 80.0, 1), (64, 64, 98.0, 3), (333, 211, 40.0, 4), (1920, 1080, 75.0, 1)):
        img = synth.g_noise(w, h, 9) if q > 90 else synth.g_struct(w, h, 9)
        bits, seg = _unstuffed_bits(oracle, img, q, mode)
        for nb in (1, 2, 3, 7, 16):
            cuts = np.sort(rng.randint(0, len(bits) + 1, nb - 1)).tolist()
            if nb >= 7:
                cuts[1] = cuts[0]                                  # an empty band
                cuts[3] = min(cuts[2] + 3, len(bits))              # a 3-bit band
                cuts = sorted(cuts)
            edges = [0] + cuts + [len(bits)]
            lens = [edges[i + 1] - edges[i] for i in range(nb)]
            stride = (max(lens) + 31) // 32 + 5
            words = np.zeros((nb, stride), np.int32)
            for i in range(nb):
                pw = _pack_words(bits[edges[i]:edges[i + 1]])
                words[i, :len(pw)] = pw
            got = engine.stitch_bands(torch.from_numpy(words).cuda(), torch.tensor(lens, dtype=torch.int64).cuda(),
                                      b"HDR", append_eoi=True)
            assert got == b"HDR" + seg + b"\xff\xd9", (w, h, q, mode, nb)


@pytest.mark.parametrize("mode", [1, 3, 4])
def test_banded_frame_equals_single_device(engine, oracle, mode):
    """Every rank's half + the root's half: P bands coded independently (sjpeg_hip_encode_band_src),
    gathered, stitched == the one-device encode == the reference."""
    for (w, h, q) in ((1920, 1080, 75.0), (640, 353, 92.0), (97, 61, 50.0)):
        img = synth.g_struct(w, h, 21)
        dev_img = torch.from_numpy(img).cuda().unsqueeze(0)
        src, _ = sj.make_source(sj.SRC_RGB, [dev_img.reshape(1, h, 3 * w)])
        tables, quant = sj.make_tables(quality=q)
        header = sj.make_header(w, h, mode, quant)
        want = oracle.encode(img, q, mode)
        nseg = sj.segment_count(w, h, mode)
        from sjpeg_amd.dist import band_ranges
        for world in (1, 2, 3, 8):
            ranges = [(b, e) for (b, e) in band_ranges(nseg, world)]
            stride = max(sj.band_bound(w, h, mode, b, e) for (b, e) in ranges if e > b)
            allw = torch.zeros((world, stride), dtype=torch.int32, device="cuda")
            alln = torch.zeros(world, dtype=torch.int64, device="cuda")
            for r, (b, e) in enumerate(ranges):
                if e > b:
                    words, nbits = engine.encode_band(src, w, h, tables, mode, b, e)
                    allw[r, :words.numel()] = words
                    alln[r] = nbits[0]
            got = engine.stitch_bands(allw, alln, header)
            assert got == want, (w, h, q, mode, world)


def test_band_argument_errors(engine):
    img = torch.zeros((1, 16, 48), dtype=torch.uint8, device="cuda")
    src, _ = sj.make_source(sj.SRC_RGB, [img])
    t, _ = sj.make_tables(quality=75)
    with pytest.raises(sj.SjpegError):
        sj.band_bound(16, 16, 1, 0, 2)                          # only one segment
    with pytest.raises(sj.SjpegError):
        engine.encode_band(src, 16, 16, t, 1, 0, 1, cap_words=8)    # buffer too small
    w = torch.zeros((2, 8), dtype=torch.int32, device="cuda")
    with pytest.raises(sj.SjpegError):
        engine.stitch_bands(w, torch.zeros(2, dtype=torch.int64, device="cuda"), b"", out_cap=16)
    torch.cuda.synchronize()


# ---- trellis quantization: methods 7 / 8 (reference src/quantize.cc:325-457) -----------------------

def test_trellis_golden_and_random(oracle, golden_small):
    n = 0
    for key, want in golden_small.items():
        img, mode, q, method = golden_input(key)
        if method == 7:
            assert sj.SjpegEncode(img, q, 7, mode) == want, key
            assert sj.SjpegEncode(img, q, 8, mode) == want, key
            n += 1
    assert n == 3
    rng = np.random.RandomState(77)
    for _ in range(24):
        w, h = int(rng.randint(1, 200)), int(rng.randint(1, 150))
        img = synth.g_struct(w, h, int(rng.randint(1 << 30))) if rng.rand() < 0.6 else \
            rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        mode = int(rng.choice([1, 3, 4]))
        q = float(rng.choice([5, 30, 60, 75, 90, 98]))
        m = int(rng.choice([7, 8]))
        assert sj.SjpegEncode(img, q, m, mode) == oracle.encode_method(img, q, mode, m), (w, h, mode, q, m)


def test_trellis_1080p(oracle):
    img = synth.g_struct(1920, 1080, 7654321)
    assert sj.SjpegEncode(img, 75.0, 7, sj.YUV_420) == oracle.encode_method(img, 75.0, 1, 7)


# ---- SJPEG_YUV_SHARP: iterative sharp RGB -> YUV 4:2:0 conversion (reference src/yuv_convert.cc) ----

def test_sharp_yuv_c1_known_answer(oracle):
    """BASELINE config #1 picture: SjpegCompress(q75) of the reference resolves to (method 4, SHARP)
    = 2571 bytes, MD5 acc8ce81...."""
    img = np.fromfile(os.path.join(ROOT, "tests", "golden", "test128.rgb"), np.uint8).reshape(128, 128, 3)
    got = sj.SjpegEncode(img, 75.0, 4, sj.YUV_SHARP)
    assert got is not None, sj.last_error()
    assert len(got) == 2571 and hashlib.md5(got).hexdigest() == "acc8ce8111f5ff4b32b3faa15ad5d994"
    assert got == oracle.encode_method(img, 75.0, 2, 4)


def test_sharp_yuv_random_vs_oracle(oracle):
    rng = np.random.RandomState(41)
    for _ in range(30):
        w, h = int(rng.randint(1, 200)), int(rng.randint(1, 160))
        kind = rng.rand()
        if kind < 0.4:
            img = synth.g_struct(w, h, int(rng.randint(1 << 30)))
        elif kind < 0.8:
            img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        else:
            img = (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)      # saturated colours: clipping sweeps
        q = float(rng.choice([30, 75, 95]))
        m = int(rng.choice([0, 4]))
        got = sj.SjpegEncode(img, q, m, sj.YUV_SHARP)
        assert got == oracle.encode_method(img, q, 2, m), (w, h, q, m)


def test_sharp_yuv_wide_and_1080p(oracle):
    for (w, h) in ((1920, 1080), (4099, 37), (5, 700)):
        img = synth.g_struct(w, h, 77)
        assert sj.SjpegEncode(img, 80.0, 0, sj.YUV_SHARP) == oracle.encode_method(img, 80.0, 2, 0), (w, h)


# ---- SJPEG_YUV_AUTO / SjpegRiskiness / SjpegCompress: need the reference's score table ---------------

@pytest.fixture(scope="module")
def risk_table():
    """The trained table lives in the reference (src/score_7.cc); tests read it out of the built
    reference (oracle/_ref travels to the GPU box) and hand it to the library at run time."""
    from oracle import refso
    if not refso.available():
        pytest.skip("oracle/_ref/libsjpeg_ref.so not built: no riskiness table to install")
    tab = refso.ref().sharpness_table()
    sj.set_riskiness_table(tab)
    return tab


def test_riskiness_matches_oracle(oracle, risk_table):
    rng = np.random.RandomState(61)
    seen = set()
    for _ in range(60):
        w, h = int(rng.randint(1, 300)), int(rng.randint(1, 200))
        k = rng.rand()
        if k < 0.3:
            img = synth.g_struct(w, h, int(rng.randint(1 << 30)))
        elif k < 0.6:
            img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        elif k < 0.8:
            img = np.repeat(rng.randint(0, 256, (h, w, 1)), 3, 2).astype(np.uint8)        # gray: 4:0:0
        else:
            img = (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
        got = sj.SjpegRiskiness(img)
        want = oracle.riskiness(img, risk_table)
        assert got == want, (w, h, got, want)
        seen.add(got[0])
    assert len(seen) >= 3                                     # the pictures really exercise several verdicts


def test_sjpeg_compress_c1_and_auto_modes(oracle, risk_table):
    """BASELINE config #1: SjpegCompress(q75) of the 128x128 test picture = 2571 bytes, MD5 acc8ce81...
    (AUTO resolves to SHARP, method 4).  Plus AUTO on pictures that resolve to the other modes."""
    img = np.fromfile(os.path.join(ROOT, "tests", "golden", "test128.rgb"), np.uint8).reshape(128, 128, 3)
    got = sj.SjpegCompress(img, 75.0)
    assert got is not None, sj.last_error()
    assert len(got) == 2571 and hashlib.md5(got).hexdigest() == "acc8ce8111f5ff4b32b3faa15ad5d994"
    rng = np.random.RandomState(62)
    pics = [synth.g_struct(211, 97, 5), rng.randint(0, 256, (64, 80, 3)).astype(np.uint8),
            np.repeat(rng.randint(0, 256, (50, 70, 1)), 3, 2).astype(np.uint8),
            (rng.randint(0, 2, (90, 60, 3)) * 255).astype(np.uint8)]
    for img in pics:
        mode, _ = oracle.riskiness(img, risk_table)
        for q, m in ((75.0, 4), (40.0, 0)):
            assert sj.SjpegEncode(img, q, m, sj.YUV_AUTO) == oracle.encode_method(img, q, mode, m), (img.shape, mode, q, m)


def test_sharp_yuv_batch_and_bgra_planes(oracle):
    """The conversion itself through the C-ABI: a batch of pictures in one call (one workgroup per
    picture), and 4-byte pixels; planes compared with the oracle's."""
    rng = np.random.RandomState(43)
    for (w, h) in ((64, 48), (101, 37), (3, 9), (640, 360)):
        imgs = [rng.randint(0, 256, (h, w, 3)).astype(np.uint8) if k % 2 else synth.g_struct(w, h, 50 + k) for k in range(5)]
        batch = torch.from_numpy(np.stack(imgs).reshape(5, h, 3 * w)).cuda()
        y, u, v = sj.sharp_yuv(sj.SRC_RGB, batch)
        bgra = np.stack([np.concatenate([im[:, :, ::-1], np.full((h, w, 1), 7, np.uint8)], 2) for im in imgs])
        y4, u4, v4 = sj.sharp_yuv(sj.SRC_BGRA, torch.from_numpy(bgra.reshape(5, h, 4 * w)).cuda())
        for k, im in enumerate(imgs):
            wy, wu, wv = oracle.sharp_yuv(im)
            assert np.array_equal(y[k].cpu().numpy(), wy) and np.array_equal(u[k].cpu().numpy(), wu) and \
                np.array_equal(v[k].cpu().numpy(), wv), (w, h, k)
            assert np.array_equal(y4[k].cpu().numpy(), wy) and np.array_equal(u4[k].cpu().numpy(), wu) and \
                np.array_equal(v4[k].cpu().numpy(), wv), (w, h, k, "bgra")


def test_sharp_sweeps_as_a_pipeline_of_workgroups(oracle):
    """The four sweeps of a picture run as four workgroups a few row pairs apart (sharp_yuv.hip, round 4), each a
    speculation on its predecessors not being the last: pictures that stop after the second sweep, after a later one,
    and never (saturated noise), short pictures (fewer row pairs than the pipeline is deep), batches that span more
    than one group of eight frames, repeated calls on one workspace."""
    rng = np.random.RandomState(44)
    def sat(h, w):
        return (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
    for (w, h, n) in ((96, 80, 19), (640, 6, 3), (33, 180, 9), (512, 300, 2)):
        imgs = []
        for k in range(n):
            kind = k % 4
            imgs.append(synth.g_struct(w, h, 60 + k) if kind == 0 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8) if kind == 1
                        else sat(h, w) if kind == 2 else np.full((h, w, 3), 40 + 9 * k, np.uint8))
        batch = torch.from_numpy(np.stack(imgs).reshape(n, h, 3 * w)).cuda()
        want = [oracle.sharp_yuv(im) for im in imgs]
        for rep in range(3):
            y, u, v = sj.sharp_yuv(sj.SRC_RGB, batch)
            for k in range(n):
                assert np.array_equal(y[k].cpu().numpy(), want[k][0]) and np.array_equal(u[k].cpu().numpy(), want[k][1]) and \
                    np.array_equal(v[k].cpu().numpy(), want[k][2]), (w, h, k, rep)


def test_sharp_sweeps_in_strips_across_the_width(oracle):
    """Round 6: a sweep is cut into strips of 192 chroma columns (a workgroup each, 32 columns of halo computed on either
    side) which meet every 32 row pairs (sharp_sweeps_strips): widths at and around the strips' edges, one to five
    strips, fewer row pairs than one block and several blocks, pictures that stop after the second sweep and ones that
    never do, batches, 4-byte pixels."""
    rng = np.random.RandomState(45)
    def sat(h, w):
        return (rng.randint(0, 2, (h, w, 3)) * 255).astype(np.uint8)
    for (w, h, n) in ((383, 66, 2), (384, 70, 3), (385, 131, 2), (386, 64, 1), (448, 258, 2), (770, 67, 3), (1153, 140, 2), (1537, 99, 1)):
        imgs = []
        for k in range(n):
            kind = (k + w) % 3
            imgs.append(synth.g_struct(w, h, 160 + k) if kind == 0 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8) if kind == 1 else sat(h, w))
        want = [oracle.sharp_yuv(im) for im in imgs]
        batch = torch.from_numpy(np.stack(imgs).reshape(n, h, 3 * w)).cuda()
        rgba = np.stack([np.concatenate([im, np.full((h, w, 1), 9, np.uint8)], 2) for im in imgs])
        for fmt, b in ((sj.SRC_RGB, batch), (sj.SRC_RGBA, torch.from_numpy(rgba.reshape(n, h, 4 * w)).cuda())):
            for rep in range(2):
                y, u, v = sj.sharp_yuv(fmt, b)
                for k in range(n):
                    assert np.array_equal(y[k].cpu().numpy(), want[k][0]) and np.array_equal(u[k].cpu().numpy(), want[k][1]) and \
                        np.array_equal(v[k].cpu().numpy(), want[k][2]), (w, h, k, fmt, rep)


def test_sharp_kernels_before_the_strips():
    """The one-workgroup-per-sweep pipeline (SJPEG_HIP_SHARP_STRIPS=0) and the in-place sweeps (SJPEG_HIP_SHARP_INPLACE=1)
    stay in the library for A/B runs: the file's sharp tests once more in processes that take them."""
    import subprocess
    import sys
    if os.environ.get("SJPEG_SHARP_INNER"):
        pytest.skip("inner run")
    for knob in ("SJPEG_HIP_SHARP_STRIPS=0", "SJPEG_HIP_SHARP_INPLACE=1"):
        k, v = knob.split("=")
        env = dict(os.environ, SJPEG_SHARP_INNER="1", **{k: v})
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-k",
                            "sharp_yuv or sharp_sweeps or sharp_and_auto"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (knob, r.stdout[-2000:], r.stderr[-1000:])


def test_adaptive_analysis_on_device_equals_host(engine):
    """AnalyseHisto's bin loops on the GPU (sjpeg_hip_adapt_sums) + the float half on the host ==
    the all-host analysis (which the CPU tests pin against the oracle), incl. min-quant limits."""
    rng = np.random.RandomState(91)
    for (w, h, mode) in ((640, 360, 1), (333, 211, 3), (97, 61, 4), (1920, 1080, 1)):
        img = synth.g_struct(w, h, 17) if w != 333 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        hist_dev = engine.scan_histogram(dev(img), mode)[0]
        hist = hist_dev.cpu().numpy().view(np.uint32)
        for q in (20.0, 75.0, 95.0):
            quant = sj.make_tables(quality=q)[1]
            for mq in (None, np.maximum(quant, 4)):
                for (dl, dc) in ((12, 1), (5, 7)):
                    t_host, q_host = sj.adapt_quant(hist, mode, quant, mq, 0x78, dl, dc)
                    t_dev, q_dev = sj.adapt_quant_device(hist_dev, mode, quant, mq, 0x78, dl, dc)
                    assert np.array_equal(q_host, q_dev), (w, h, mode, q, dl, dc)
                    assert bytes(t_host) == bytes(t_dev)


def test_adaptive_decision_on_device_equals_host(engine):
    """The float half of AnalyseHisto on the GPU (sjpeg_hip_adapt_decide: line fits, lambda, step per position -- what
    sjpeg_hip_encode_batch_src runs) == the host's (which the CPU tests pin against the oracle): pictures, and synthetic
    histograms that sit on its branches -- sparse positions, flat and single-bin ones, huge counts, steps cut off by
    min_quant / 255, every qdelta_max, 4:0:0."""
    rng = np.random.RandomState(1234)
    hists = []
    for (w, h, mode) in ((640, 360, 1), (333, 211, 3), (97, 61, 4), (1920, 1080, 1)):
        img = synth.g_struct(w, h, 17) if w != 333 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        hists.append((engine.scan_histogram(dev(img), mode)[0].cpu().numpy().view(np.uint32).reshape(2, 64, 128), mode))
    for k in range(40):                                   # synthetic: geometric tails of random scale and population
        hs = np.zeros((2, 64, 128), np.uint32)
        for t in range(2):
            for pos in range(64):
                kind = rng.randint(6)
                n = int(rng.choice([0, 3, 50, 5000, 400000, 20000000]))
                if kind == 0 or n == 0:
                    continue
                if kind == 1:
                    hs[t, pos, rng.randint(128)] = n
                elif kind == 2:
                    hs[t, pos, :] = n // 128
                else:
                    scale = float(rng.choice([0.3, 2.0, 9.0, 40.0]))
                    bins = np.minimum(rng.geometric(1.0 / (1.0 + scale), size=min(n, 20000)) - 1, 140)
                    cnt = np.bincount(bins[bins < 128], minlength=128).astype(np.uint64) * max(n // min(n, 20000), 1)
                    hs[t, pos, :] = np.minimum(cnt, 0xffffffff).astype(np.uint32)
        hists.append((hs, int(rng.choice([1, 3, 4]))))
    checked = 0
    for hs, mode in hists:
        hd = torch.from_numpy(hs.view(np.int32).copy()).cuda().unsqueeze(0)
        for q in (8.0, 50.0, 75.0, 97.0):
            quant = sj.make_tables(quality=q)[1]
            for mq in (None, np.maximum(quant, 4), np.minimum(quant.astype(np.int32) + 3, 255).astype(np.uint8)):
                for (dl, dc) in ((12, 1), (5, 7), (0, 0), (-3, 12)):
                    _, q_host = sj.adapt_quant(hs, mode, quant, mq, 0x78, dl, dc)
                    q_dev = sj.adapt_quant_on_device(hd, mode, quant, mq, dl, dc)[0]
                    ntab = 1 if mode == 4 else 2
                    want = np.asarray(q_host, np.uint8).reshape(2, 64)
                    # (the host clamps to min_quant when it finalizes; the device result is the step choice itself)
                    t_dev = sj.ScanTables()
                    qd = q_dev.copy()
                    sj.lib().sjpeg_hip_finalize_quant(qd.ctypes.data, None if mq is None else np.ascontiguousarray(mq, np.uint8).ctypes.data, 0x78, C.byref(t_dev))
                    assert np.array_equal(qd[:ntab], want[:ntab]), (mode, q, dl, dc, checked)
                    checked += 1
    assert checked == 44 * 4 * 3 * 4


def test_keep_and_replay_flags(oracle):
    """SJPEG_HIP_QUANT_KEEP / REPLAY: the statistics pass leaves its quantized blocks behind and the
    encode pass entropy-codes them without touching the pixels again (plain and trellis)."""
    eng = sj.Engine(0)
    for (w, h, mode, q) in ((333, 211, 1, 75.0), (640, 360, 3, 90.0), (97, 61, 4, 40.0)):
        img = synth.g_struct(w, h, 4)
        frames = dev(img)
        for trellis in (False, True):
            tables, quant = sj.make_tables(quality=q)
            if trellis:
                tables.flags = sj.QUANT_TRELLIS
                for c in range(2):
                    for i in range(256):
                        tables.trellis_len[c][i] = tables.ac_codes[c][i] & 0xff
            header = sj.make_header(w, h, mode, quant)
            plain = eng.encode_frames(frames, tables, header, mode)
            want = bytes(plain[0][0, :int(plain[1][0].item())].cpu().numpy())
            if not trellis:
                assert want == oracle.encode(img, q, mode)
            tables.flags |= sj.QUANT_KEEP
            eng.scan_symbol_stats(frames, tables, mode)
            tables.flags = (tables.flags & ~sj.QUANT_KEEP) | sj.QUANT_REPLAY
            out, sizes = eng.encode_frames(torch.zeros_like(frames), tables, header, mode)   # pixels are not read
            assert bytes(out[0, :int(sizes[0].item())].cpu().numpy()) == want, (w, h, mode, trellis)
    fresh = sj.Engine(0)
    tables, quant = sj.make_tables(quality=75.0)
    tables.flags = sj.QUANT_REPLAY
    with pytest.raises(sj.SjpegError):
        fresh.encode_frames(dev(synth.g_struct(32, 32, 1)), tables, b"", 1)
    torch.cuda.synchronize()


def test_sharp_and_auto_with_padded_and_bottom_up_rows(oracle, risk_table):
    """Row padding must not leak and a bottom-up picture equals the flipped one, also through the
    sharp conversion and the riskiness stencil (reference tests/unit_test.cc:246-342 idea)."""
    img = synth.g_struct(61, 45, 15)
    lib = sj.lib()
    for mode in (sj.YUV_SHARP, sj.YUV_AUTO):
        want = sj.SjpegEncode(img, 80.0, 4, mode)
        eff = 2 if mode == sj.YUV_SHARP else oracle.riskiness(img, risk_table)[0]
        assert want == oracle.encode_method(img, 80.0, eff, 4)
        padded = np.full((45, 61 * 3 + 13), 0x5C, np.uint8)
        padded[:, :183] = img.reshape(45, 183)
        out = C.POINTER(C.c_uint8)()
        n = lib.SjpegEncode(padded.ctypes.data, 61, 45, padded.strides[0], C.byref(out), C.c_float(80.0), 4, mode)
        assert C.string_at(out, n) == want
        lib.SjpegFreeBuffer(out)
        flipped = img[::-1].copy()
        assert sj.SjpegEncode(flipped, 80.0, 4, mode, stride=-flipped.strides[0]) == want


# ---- exchange step of the multi-device batch path ------------------------------------------------

def test_compact_streams_vs_restatement(engine):
    """sjpeg_hip_compact_streams against the torch restatement the gloo tests use
    (tests/test_dist_cpu.py::_compact_torch): ragged sizes, a frame that fills its slot, a batch
    that does not fit the packed buffer, and the streams of a real encode call."""
    from test_dist_cpu import _compact_torch
    rng = np.random.default_rng(5)
    stride = 4096 + 16
    for n in (1, 2, 7, 64, 300):
        out = torch.from_numpy(rng.integers(0, 256, (n, stride), dtype=np.uint8)).cuda()
        sz = rng.integers(1, stride + 1, n)
        sz[0] = stride                                   # full slot
        if n > 2:
            sz[1], sz[2] = 1, 16
        sizes = torch.from_numpy(sz.astype(np.int64)).cuda()
        need = int(((sz + 15) & ~15).sum())
        packed, offs = sj.compact_streams(out, sizes, n, need)
        want, woffs = _compact_torch(out.cpu(), sizes.cpu(), n, need)
        assert offs.cpu().tolist() == woffs.tolist() and int(offs[-1]) == need
        assert torch.equal(packed.cpu(), want)
        # too small a buffer: the needed size is still reported, the frames that fit are intact
        if n > 2:
            small = int(woffs[n - 1])                    # room for all but the last frame
            p2, o2 = sj.compact_streams(out, sizes, n, small)
            # (bit 63 = SJPEG_HIP_PACKED_OVERFLOW: the batch did not fit; the low bits are still the bytes needed)
            assert int(o2[-1]) < 0 and int(o2[-1]) & ((1 << 63) - 1) == need and torch.equal(p2.cpu(), want[:small])
    # the real thing: a batch of coded frames survives the packing
    imgs = [synth.g_struct(160, 96, 70 + k) for k in range(5)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    tables, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(160, 96, sj.YUV_420, quant)
    ostride = (sj.frame_bound(160, 96, sj.YUV_420, len(header)) + 15) & ~15
    out, sizes = engine.encode_frames(frames, tables, header, sj.YUV_420, out_stride=ostride)
    packed, offs = sj.compact_streams(out, sizes)
    torch.cuda.synchronize()
    o, s, p = offs.cpu().numpy(), sizes.cpu().numpy(), packed.cpu().numpy()
    for k in range(5):
        assert p[o[k]:o[k] + s[k]].tobytes() == bytes(out[k, :int(s[k])].cpu().numpy())
    # misaligned arguments are refused, not mis-copied
    with pytest.raises(sj.SjpegError):
        sj.compact_streams(out[:, 1:], sizes)


def test_gather_streams_c_abi_one_rank(engine, oracle):
    """sjpeg_hip_comm_create + sjpeg_hip_gather_rows / _bytes (the RCCL exchange of the C-ABI) with a
    communicator of ONE rank: all-gather of the row, the host read, the root's own block copied into the
    gathered buffer -- with the DEFAULT out_stride of encode_frames (frame_bound: a multiple of 16), and
    a frame that does not fit its slot refused before anything moves."""
    w, h, nf = 200, 120, 5
    tables, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    imgs = [synth.g_struct(w, h, 900 + k) for k in range(nf)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    out, sizes = engine.encode_frames(frames, tables, header, sj.YUV_420)        # default stride
    assert out.stride(0) % 16 == 0
    comm = sj.Comm(sj.comm_unique_id(), 0, 1)
    try:
        per_max = nf + 2                                  # (a rank may hold fewer frames than the largest)
        packed, offsets = sj.compact_streams(out, sizes)
        rows_dev = torch.zeros(2 * (per_max + 2), dtype=torch.int64, device="cuda")
        rows, offs = comm.gather_rows(offsets, sizes, nf, per_max, rows_dev)
        assert int(rows[0][1]) == nf and int(offs[1]) == int(rows[0][0]) == int(offsets[nf])
        gathered = torch.zeros(int(offs[1]), dtype=torch.uint8, device="cuda")
        comm.gather_bytes(0, packed, per_max, rows, offs, gathered)
        torch.cuda.synchronize()
        host, o = gathered.cpu().numpy(), 0
        for k in range(nf):
            n = int(rows[0][2 + k])
            assert host[o:o + n].tobytes() == oracle.encode(imgs[k], 75.0, 1), k
            o += (n + 15) & ~15
        small = torch.empty(int(gathered.numel()) - 16, dtype=torch.uint8, device="cuda")
        with pytest.raises(sj.SjpegError):
            comm.gather_bytes(0, packed, per_max, rows, offs, small)             # the root's buffer is too small
        # a frame that did not fit its slot (size 0): refused by the rows on every rank
        tight = (len(header) + 2 + 64 + 15) & ~15
        out2, sizes2 = engine.encode_frames(frames, tables, header, sj.YUV_420, out_stride=tight)
        torch.cuda.synchronize()
        assert int(sizes2.sum()) == 0
        packed2, offsets2 = sj.compact_streams(out2, sizes2)
        with pytest.raises(sj.SjpegError):
            comm.gather_rows(offsets2, sizes2, nf, per_max, rows_dev)
    finally:
        comm.close()


def test_frame_tensor_layout_is_checked(engine):
    """A permuted / channel-first / sliced view must be refused (the C-ABI sees two strides only)."""
    img = torch.from_numpy(synth.g_struct(64, 48, 3)).cuda()
    tables, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(64, 48, sj.YUV_420, quant)
    chw = img.permute(2, 0, 1).contiguous().permute(1, 2, 0).unsqueeze(0)     # [1, H, W, 3] view of CHW data
    with pytest.raises(sj.SjpegError):
        engine.encode_frames(chw, tables, header, sj.YUV_420)
    with pytest.raises(sj.SjpegError):
        engine.encode_frames(img.unsqueeze(0)[:, :, ::2, :], tables, header, sj.YUV_420)
    out, sizes = engine.encode_frames(chw.contiguous(), tables, header, sj.YUV_420)
    assert int(sizes[0]) > 0


def test_overlapped_steps_on_streams(engine):
    """The stream / event order of sjpeg_amd.dist.overlapped_steps (bench.py's multi-rank loop) on one
    device: step s codes picture set s into buffer set s & 1 while the streams of step s - 1 are
    packed on a side stream; every step's packed streams must be that step's."""
    from sjpeg_amd.dist import overlapped_steps
    w, h, nf, nsteps = 320, 192, 6, 7
    tables, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    stride = (sj.frame_bound(w, h, sj.YUV_420, len(header)) + 15) & ~15
    sets = [torch.from_numpy(np.stack([synth.g_struct(w, h, 500 + 10 * s + k) for k in range(nf)])).cuda() for s in range(nsteps)]
    outs = [torch.empty((nf, stride), dtype=torch.uint8, device="cuda") for _ in range(2)]
    sizes = [torch.zeros(nf, dtype=torch.int64, device="cuda") for _ in range(2)]
    state = {"s": 0}

    def encode(b):
        engine.encode_frames(sets[state["s"]], tables, header, sj.YUV_420, out=outs[b], sizes=sizes[b], out_stride=stride)
        state["s"] += 1

    def exchange(b):
        packed, offs = sj.compact_streams(outs[b], sizes[b])
        return packed.clone(), offs.clone(), sizes[b].clone()

    got = overlapped_steps(nsteps, encode, exchange, use_streams=True)
    torch.cuda.synchronize()
    assert len(got) == nsteps
    ref = sj.Engine(0)
    for s, (packed, offs, sz) in enumerate(got):
        want_out, want_sz = ref.encode_frames(sets[s], tables, header, sj.YUV_420, out_stride=stride)
        torch.cuda.synchronize()
        p, o, n = packed.cpu().numpy(), offs.cpu().numpy(), sz.cpu().numpy()
        assert n.tolist() == want_sz.cpu().numpy().tolist(), s
        for k in range(nf):
            assert p[o[k]:o[k] + n[k]].tobytes() == bytes(want_out[k, :int(n[k])].cpu().numpy()), (s, k)


def test_scratch_sized_from_out_stride(engine, oracle):
    """The engine sizes its segment scratch from out_stride, not for the worst case: with a slot that is
    just big enough, busy segments run over their segment slot into the frame's pool and the checked
    walk takes its rows there -- same bytes as with worst-case scratch; one byte less and the frame
    reports size 0 (never a truncated stream); a batch mixes frames that fit and frames that do not."""
    cases = [(synth.g_noise(400, 304, 11), 99.0, 3), (synth.g_noise(640, 360, 12), 100.0, 1),
             (synth.g_struct(1280, 720, 13), 75.0, 1), (synth.g_noise(97, 61, 14), 100.0, 4)]
    for img, q, mode in cases:
        h, w = img.shape[:2]
        want = oracle.encode(img, q, mode)
        t, quant = sj.make_tables(quality=q)
        header = sj.make_header(w, h, mode, quant)
        for slack in (0, 16, 4096):
            stride = len(want) + slack
            out, sizes = engine.encode_frames(dev(img), t, header, mode, out_stride=stride)
            torch.cuda.synchronize()
            assert int(sizes[0]) == len(want) and bytes(out[0, :len(want)].cpu().numpy()) == want, (w, h, q, slack)
        out, sizes = engine.encode_frames(dev(img), t, header, mode, out_stride=len(want) - 1)
        torch.cuda.synchronize()
        assert int(sizes[0]) == 0
    # a batch in which only the busy frames overrun the common slot
    calm, busy = synth.g_struct(320, 240, 21), synth.g_noise(320, 240, 22)
    t, quant = sj.make_tables(quality=97.0)
    header = sj.make_header(320, 240, 1, quant)
    want = [oracle.encode(calm, 97.0, 1), oracle.encode(busy, 97.0, 1)]
    assert len(want[1]) > len(want[0]) + 64
    stride = (len(want[0]) + 64 + 15) & ~15
    frames = torch.from_numpy(np.stack([calm, busy, calm, busy, busy, calm])).cuda()
    out, sizes = engine.encode_frames(frames, t, header, 1, out_stride=stride)
    torch.cuda.synchronize()
    sz = sizes.cpu().numpy().tolist()
    assert sz == [len(want[0]), 0, len(want[0]), 0, 0, len(want[0])]
    for k in (0, 2, 5):
        assert bytes(out[k, :sz[k]].cpu().numpy()) == want[0]


def test_replayed_blocks_are_classified_by_the_coding_tables(oracle):
    """Methods 7 / 8 quantize in the statistics pass and replay the kept blocks in the encode pass, which
    codes them with OPTIMISED tables: whether a block may take the lean walk depends on those tables,
    not on the ones the statistics pass ran with (found by tools/gpu_soak.py: q >= 97, trellis)."""
    rng = np.random.RandomState(77)
    for (w, h, q, method, mode) in ((215, 279, 100.0, 7, 1), (331, 257, 97.0, 8, 1), (260, 190, 100.0, 8, 3),
                                   (300, 200, 100.0, 4, 1), (180, 260, 97.0, 1, 3)):
        for img in (rng.randint(0, 256, (h, w, 3)).astype(np.uint8), synth.g_struct(w, h, 5 + w)):
            got = sj.SjpegEncode(img, q, method, mode)
            assert got == oracle.encode_method(img, q, mode, method), (w, h, q, method, mode)
    # the batch path replays too (sjpeg_hip_encode_batch_src, default parameters = method 4)
    frames = np.stack([rng.randint(0, 256, (120, 160, 3)).astype(np.uint8) for _ in range(3)])
    got = sj.encode_device_method(torch.from_numpy(frames).cuda(), 100.0, 1, 4)
    for k in range(3):
        assert got[k] == oracle.encode_method(frames[k], 100.0, 1, 4), k


# ---- optional restart-marker mode (never the reference's bytes: same coefficients, same pixels) ----

def test_restart_mode_equals_restatement_and_decodes_to_the_same_pixels(engine, oracle):
    """SJPEG_HIP_RESTART_MARKERS: every engine segment a restart interval.  The bytes are compared with the
    oracle's restatement of the standard's rule (oracle/sjpeg_oracle.c orc_encode_rst: not something the
    reference writes) and, through an independent decoder, the pixels with those of the exact stream."""
    Image = pytest.importorskip("PIL.Image")
    cases = [(640, 360, 75.0, 1), (333, 211, 90.0, 3), (300, 260, 50.0, 4), (1920, 1080, 75.0, 1),
             (17, 13, 75.0, 1), (16 * 41, 16, 60.0, 1), (16 * 41 + 1, 16, 60.0, 1), (700, 500, 99.0, 1)]
    for (w, h, q, mode) in cases:
        img = synth.g_noise(w, h, 3) if q > 95 else synth.g_struct(w, h, 9 + w)
        t, quant = sj.make_tables(quality=q)
        t.flags |= sj.RESTART_MARKERS
        header = sj.header_add_restart(sj.make_header(w, h, mode, quant), mode)
        out, sizes = engine.encode_frames(dev(img), t, header, mode)
        torch.cuda.synchronize()
        got = bytes(out[0, :int(sizes[0])].cpu().numpy())
        want = oracle.encode_rst(img, q, mode, sj.restart_interval(mode))
        assert got == want, (w, h, q, mode, len(got), len(want))
        exact = oracle.encode(img, q, mode)
        a = np.asarray(Image.open(io.BytesIO(got)).convert("RGB"))
        b = np.asarray(Image.open(io.BytesIO(exact)).convert("RGB"))
        assert np.array_equal(a, b), (w, h, q, mode)
    # a batch, and the exact mode right after on the same engine (the flag must not stick)
    frames = np.stack([synth.g_struct(320, 240, 70 + k) for k in range(4)])
    t, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(320, 240, 1, quant)
    t.flags |= sj.RESTART_MARKERS
    out, sizes = engine.encode_frames(torch.from_numpy(frames).cuda(), t, sj.header_add_restart(header, 1), 1)
    torch.cuda.synchronize()
    for k in range(4):
        assert bytes(out[k, :int(sizes[k])].cpu().numpy()) == oracle.encode_rst(frames[k], 75.0, 1, 41)
    t.flags &= ~sj.RESTART_MARKERS
    out, sizes = engine.encode_frames(torch.from_numpy(frames).cuda(), t, header, 1)
    torch.cuda.synchronize()
    for k in range(4):
        assert bytes(out[k, :int(sizes[k])].cpu().numpy()) == oracle.encode(frames[k], 75.0, 1)


def test_restart_bands_concatenate_to_the_one_device_stream(engine, oracle):
    """The north star's literal form: a frame cut at restart markers into bands (one per device), every band
    coded on its own (sjpeg_hip_encode_intervals_src), the bytes concatenated behind the header.  P bands
    simulated on one GPU; equal to the one-device restart stream and to the oracle's restatement."""
    for (w, h, q, mode) in ((1920, 1080, 75.0, 1), (640, 360, 90.0, 3), (333, 211, 60.0, 4), (4000, 48, 97.0, 1)):
        img = synth.g_noise(w, h, 4) if q > 95 else synth.g_struct(w, h, 3 + h)
        t, quant = sj.make_tables(quality=q)
        t.flags |= sj.RESTART_MARKERS
        header = sj.header_add_restart(sj.make_header(w, h, mode, quant), mode)
        want = oracle.encode_rst(img, q, mode, sj.restart_interval(mode))
        nseg = sj.segment_count(w, h, mode)
        rows = dev(img).reshape(1, h, 3 * w)             # (kept alive: the source holds raw pointers)
        src, _ = sj.make_source(sj.SRC_RGB, [rows])
        from sjpeg_amd.dist import band_ranges
        for P in (1, 2, 3, 8):
            parts = []
            for (b, e) in band_ranges(nseg, P):
                if e > b:
                    out, size = engine.encode_intervals(src, w, h, t, mode, b, e)
                    torch.cuda.synchronize()
                    assert int(size[0]) > 0
                    parts.append(bytes(out[:int(size[0])].cpu().numpy()))
            assert header + b"".join(parts) + b"\xff\xd9" == want, (w, h, q, mode, P)


def test_host_api_codes_again_when_the_first_capacity_is_too_small(oracle):
    """The host API sizes its output buffer for half a byte per sample and repeats a frame against the
    worst-case bound if that was too small: dense pictures (noise at q 100: 3-5 bytes per pixel) give the
    reference's bytes through the second pass, for the plain encode and for the size-search pass."""
    for (w, h, q, mode) in ((512, 512, 100.0, 3), (640, 400, 100.0, 1), (333, 517, 99.0, 4)):
        img = synth.g_noise(w, h, 100 + w)
        want = oracle.encode(img, q, mode)
        samples = w * h * (3 if mode == 3 else 1 if mode == 4 else 1.5)
        assert len(want) > 65536 + 1024 + samples / 2, "picture too calm to need the second pass"
        assert sj.SjpegEncode(img, q, 0, mode) == want, (w, h, q, mode)
        # ... and a calm picture right after it, through the buffers the dense one left behind
        calm = synth.g_struct(w, h, 3)
        assert sj.SjpegEncode(calm, 75.0, 0, mode) == oracle.encode(calm, 75.0, mode)


def test_trim_gives_scratch_back_and_the_next_call_allocates_again(engine, oracle):
    img = synth.g_struct(1024, 768, 9)
    want = oracle.encode(img, 80.0, 1)
    t, quant = sj.make_tables(quality=80.0)
    header = sj.make_header(1024, 768, 1, quant)
    out, sizes = engine.encode_frames(dev(img), t, header, 1)
    torch.cuda.synchronize()
    before = engine.scratch_bytes()
    engine.trim()
    after = engine.scratch_bytes()
    assert before > 1 << 20 and after < 1 << 16, (before, after)
    for _ in range(2):
        out, sizes = engine.encode_frames(dev(img), t, header, 1)
        torch.cuda.synchronize()
        assert bytes(out[0, :int(sizes[0])].cpu().numpy()) == want
    # pipelined mode keeps two sets: both go, both come back
    engine.set_pipelined(True)
    try:
        outs = [engine.encode_frames(dev(img), t, header, 1) for _ in range(3)]
        engine.wait()
        torch.cuda.synchronize()
        engine.trim()
        assert engine.scratch_bytes() < 1 << 16
        outs = [engine.encode_frames(dev(img), t, header, 1) for _ in range(3)]
        engine.wait()
        torch.cuda.synchronize()
        for out, sizes in outs:
            assert bytes(out[0, :int(sizes[0])].cpu().numpy()) == want
    finally:
        engine.set_pipelined(False)
    # the host API's per-thread cache
    assert sj.SjpegEncode(img, 80.0, 0, 1) == want
    assert sj.host_trim() > 1 << 20
    assert sj.host_trim() < 1 << 16
    assert sj.SjpegEncode(img, 80.0, 0, 1) == want


def _seg_row(mode):
    """(width, MCU height) of a picture whose MCU rows are exactly one K1 segment each (41 / 82 / 246 MCUs)"""
    px = 16 if mode == 1 else 8
    return sj.restart_interval(mode) * px, px


def _noise_rows(mode, amps, seed):
    w, mh = _seg_row(mode)
    rng = np.random.RandomState(seed)
    amp = np.asarray(amps, np.float64).repeat(mh)[:, None, None]
    return np.clip(128 + rng.randint(-128, 128, (mh * len(amps), w, 3)) * amp / 128.0, 0, 255).astype(np.uint8)


def _amp_for_row_bytes(oracle, mode, q, nbytes):
    """Noise amplitude at which one MCU row (= one segment) codes to about nbytes of entropy data."""
    empty = len(oracle.encode(_noise_rows(mode, [0.0], 1), q, mode))
    lo, hi = 0.0, 128.0
    for _ in range(12):
        mid = (lo + hi) / 2
        if len(oracle.encode(_noise_rows(mode, [mid], 1), q, mode)) - empty < nbytes:
            lo = mid
        else:
            hi = mid
    return (lo + hi) / 2


def _sweep_segment_lengths(engine, oracle, mode, q, words_lo, words_hi, nseg, per_seg_strides, seed):
    """One segment per MCU row, noise whose amplitude grows down the picture: the segments' lengths
    sweep every word count from words_lo to words_hi, with every bit alignment; out_stride (per
    segment) chooses the slot size, i.e. where a segment stops fitting its slot and continues in the pool."""
    a_lo, a_hi = (_amp_for_row_bytes(oracle, mode, q, 4 * n) for n in (words_lo, words_hi))
    assert a_lo < a_hi < 127.0, (a_lo, a_hi)
    img = _noise_rows(mode, np.linspace(a_lo, a_hi, nseg), seed)
    h, w = img.shape[:2]
    want = oracle.encode(img, q, mode)
    assert 3.6 * words_lo * nseg < len(want) < 4.4 * words_hi * nseg, len(want) / nseg
    t, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, mode, quant)
    d = dev(img)

    def check(out, sizes, what):
        got = out[0, :int(sizes[0])].cpu().numpy()
        assert len(got) == len(want), what
        diff = np.nonzero(got != np.frombuffer(want, np.uint8))[0]
        assert len(diff) == 0, (what, len(diff), int(diff[0]))

    for per_seg in per_seg_strides:
        out, sizes = engine.encode_frames(d, t, header, mode, out_stride=per_seg * nseg)
        torch.cuda.synchronize()
        check(out, sizes, (mode, q, per_seg))
    # the second buffer set of the pipelined mode, with its own pool
    engine.set_pipelined(True)
    try:
        outs = [engine.encode_frames(d, t, header, mode, out_stride=per_seg_strides[0] * nseg) for _ in range(3)]
        engine.wait()
        torch.cuda.synchronize()
        for k, (out, sizes) in enumerate(outs):
            check(out, sizes, (mode, q, "pipelined", k))
    finally:
        engine.set_pipelined(False)


def test_segments_of_every_length_around_the_slot_size(engine, oracle):
    """Segment lengths from ~100 words below to ~200 above a 1024-word slot.  Slots of 1024 / 1088 / 1152
    words (chosen through out_stride: a slot is three quarters of a segment's share, 1024 words at least) put the boundary between 'fits the slot' and 'continues in the
    pool' at three places of that sweep.  Found at 65535 x 65535 (tools/max_frame_check.py): a segment
    of slot_words - 3 words had its last word placed from the wrong source words."""
    _sweep_segment_lengths(engine, oracle, 1, 90.0, 940, 1230, 2400, (5000, 5760, 6100, 20000), 2024)
    _sweep_segment_lengths(engine, oracle, 3, 90.0, 940, 1230, 1600, (5000, 5760, 6100, 20000), 2025)
    _sweep_segment_lengths(engine, oracle, 4, 85.0, 940, 1230, 1600, (5000, 5760, 6100, 20000), 2026)


def test_segments_of_every_length_around_the_stitch_window(engine, oracle):
    """K1 stitches a segment through an 8 KiB window in LDS, in several rounds if it is longer: lengths
    sweeping across one window (q 97: lean and checked parts mixed) and across two (q 100), with slots
    that end before, inside and behind the window boundary."""
    _sweep_segment_lengths(engine, oracle, 1, 97.0, 1900, 2500, 1500, (9400, 11500, 40000), 7)
    _sweep_segment_lengths(engine, oracle, 1, 100.0, 3800, 4500, 1500, (17800, 22400, 80000), 8)
    _sweep_segment_lengths(engine, oracle, 3, 97.0, 1900, 2500, 1000, (9400, 11500, 40000), 9)


# ---- the lanes of the packed fDCT at their proven extremes (tools/int16_ranges.py) --------------

_PAIRS = {"blue|yellow (Cb +128 / -127)": ((0, 0, 255), (255, 255, 0)),
          "red|cyan (Cr +128 / -127)": ((255, 0, 0), (0, 255, 255)),
          "white|black (Y +127 / -128)": ((255, 255, 255), (0, 0, 0))}


def _extremal_masks():
    import json
    pats = json.load(open(os.path.join(ROOT, "tests", "golden", "extremal_patterns.json")))["patterns"]
    masks = sorted({int(p[k], 16) for p in pats for k in ("max", "min")})
    assert len(masks) >= 40
    return masks


def _paint(masks, scale, hi, lo, per_row=12):
    """One tile of 8 x 8 cells (`scale` pixels a side) per mask: cell i = `hi` where bit i is set, else `lo`."""
    side = 8 * scale
    rows = (len(masks) + per_row - 1) // per_row
    img = np.zeros((rows * side, per_row * side, 3), np.uint8)
    img[:] = np.array(lo, np.uint8)
    for t, m in enumerate(masks):
        bits = np.array([(m >> i) & 1 for i in range(64)], bool).reshape(8, 8)
        cell = np.where(bits[..., None], np.array(hi, np.uint8), np.array(lo, np.uint8)).astype(np.uint8)
        y0, x0 = (t // per_row) * side, (t % per_row) * side
        img[y0:y0 + side, x0:x0 + side] = cell.repeat(scale, 0).repeat(scale, 1)
    return img


def _tap_and_bytes(engine, oracle, img, mode, q, what):
    t, quant = sj.make_tables(quality=q)
    zz = engine.scan_coeffs(dev(img), t, mode)
    torch.cuda.synchronize()
    want = oracle.scan_coeffs(img, quant, 0x78, mode)
    assert (zz[0].cpu().numpy() == want).all(), ("tap", what, mode, q)
    got = sj.encode_device(dev(img), q, mode, engine=engine)[0]
    assert got == oracle.encode(img, q, mode), ("bytes", what, mode, q)


def test_extremal_patterns_of_the_packed_fdct(engine, oracle):
    """The corner patterns at which the int16 lanes / 32-bit accumulators of fdct_col8_pk, fdct_row8_pk and row_quant
    come closest to their type (c1 = a0 - a3 reaches 32 640 of 32 767), painted with the colour pairs that put Cb, Cr
    and Y at the ends of their ranges: coefficients (tap) and bytes against the oracle.  4:4:4 and 4:0:0 see the
    pattern per pixel; 4:2:0 sees it per pixel in luma and -- at two pixels per cell -- exactly in chroma."""
    masks = _extremal_masks()
    for name, (hi, lo) in _PAIRS.items():
        for scale in (1, 2):
            img = _paint(masks, scale, hi, lo)
            for mode in (3, 1, 4):
                for q in (100.0, 75.0):
                    _tap_and_bytes(engine, oracle, img, mode, q, (name, scale))
            # the same corners with the two colours exchanged (the minimum of every form)
            _tap_and_bytes(engine, oracle, _paint(masks, scale, lo, hi), 3, 100.0, (name, scale, "exchanged"))
            _tap_and_bytes(engine, oracle, _paint(masks, scale, lo, hi), 1, 100.0, (name, scale, "exchanged"))


def test_extreme_stripes_through_tap_and_bytes(engine, oracle):
    """Column and row stripes of period 1 / 2 / 4 / 8 (and 16 for the 4:2:0 chroma blocks) between the colours
    that put a component at both ends of its range, every phase against the block grid."""
    for name, (hi, lo) in _PAIRS.items():
        for period in (1, 2, 4, 8, 16):
            for phase in range(0, min(period, 8), max(1, period // 4)):
                x = np.arange(160)
                on = (((x + phase) // period) & 1).astype(bool)
                col = np.where(on[None, :, None], np.array(hi, np.uint8), np.array(lo, np.uint8)).astype(np.uint8)
                for img in (np.broadcast_to(col, (96, 160, 3)).copy(),
                            np.broadcast_to(col.transpose(1, 0, 2), (160, 96, 3)).copy()):
                    for mode in (3, 1, 4):
                        _tap_and_bytes(engine, oracle, img, mode, 100.0 if period < 4 else 90.0, (name, period, phase))


def test_tap_on_lattices_of_constant_columns_and_rows(engine, oracle):
    """Blocks whose columns (rows) are constant, drawn from the corners of the RGB cube: the column pass feeds the
    row pass one non-zero row (column) of the largest magnitudes -- what g_struct / g_noise never do."""
    rng = np.random.RandomState(404)
    corners = np.array([[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)], np.uint8)
    for trial in range(6):
        for cell in (1, 2):
            w, h = 192, 128
            cx = corners[rng.randint(0, 8, w // cell)].repeat(cell, 0)[:w]          # one colour per column
            cy = corners[rng.randint(0, 8, h // cell)].repeat(cell, 0)[:h]          # one colour per row
            imgs = [np.broadcast_to(cx[None], (h, w, 3)).copy(), np.broadcast_to(cy[:, None], (h, w, 3)).copy()]
            mix = np.where((rng.randint(0, 2, (h // 8, w // 8)).repeat(8, 0).repeat(8, 1))[..., None] == 1, imgs[0], imgs[1])
            imgs.append(mix.astype(np.uint8))                                       # column blocks beside row blocks
            for img in imgs:
                for mode in (3, 1, 4):
                    _tap_and_bytes(engine, oracle, img, mode, (100.0, 95.0, 60.0)[trial % 3], (trial, cell))


def test_every_block_makes_four_parts(engine, oracle):
    """Noise at high quality: every quarter of every block holds a non-zero level, so a segment makes the largest
    number of parts K1's sorted part list can be asked to hold (4 x 246 coded blocks, every mode).  Round 4: with
    83 MCUs per 4:4:4 segment the list ran 24 bytes into the bins of the counting sort while other waves were still
    reading them -- wrong in 1 run of 2 at 1080p, never in the small pictures."""
    rng = np.random.RandomState(77)
    img = rng.randint(0, 256, (1080, 1920, 3)).astype(np.uint8)
    for mode in (3, 4, 1):
        want = oracle.encode(img, 92.0, mode)
        d = dev(img)
        for rep in range(8):
            got = sj.encode_device(d, 92.0, mode, engine=engine)[0]
            assert got == want, (mode, rep)


def test_packed_output_equals_the_strided_frames(engine, oracle):
    """sjpeg_hip_encode_scan_packed_src: the frames back to back at multiples of 16 (zero padding), offsets = the
    running sum of the 16-aligned sizes -- what sjpeg_hip_compact_streams makes of the strided batch, byte for
    byte --; frames of different sizes, restart mode, a frame that does not fit its share (size 0, no room)."""
    rng = np.random.RandomState(5)
    w, h, nf = 200, 120, 7
    imgs = [synth.g_struct(w, h, 70 + k) if k % 3 else rng.randint(0, 256, (h, w, 3)).astype(np.uint8) for k in range(nf)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    for q, flags in ((75.0, 0), (95.0, 0), (60.0, sj.RESTART_MARKERS)):
        tables, quant = sj.make_tables(quality=q)
        tables.flags = flags
        header = sj.make_header(w, h, sj.YUV_420, quant)
        if flags:
            header = sj.header_add_restart(header, sj.YUV_420)
        out, sizes = engine.encode_frames(frames, tables, header, sj.YUV_420)
        stride = int(out.stride(0))
        want_packed, want_offsets = sj.compact_streams(out, sizes)
        flat = torch.full((nf * stride + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        sizes2 = torch.zeros(nf, dtype=torch.int64, device="cuda")
        offs2 = torch.zeros(nf + 1, dtype=torch.int64, device="cuda")
        engine.encode_frames_packed(frames, tables, header, sj.YUV_420, flat, sizes2, offs2, stride)
        torch.cuda.synchronize()
        assert torch.equal(sizes, sizes2) and torch.equal(offs2, want_offsets)
        total = int(offs2[nf])
        assert torch.equal(flat[:total], want_packed[:total])
        assert int((flat[total:] != 0xA5).sum()) == 0                       # nothing behind the batch is touched
        if not flags:
            o = offs2.cpu().numpy()
            for k in range(nf):
                assert flat[int(o[k]):int(o[k]) + int(sizes2[k])].cpu().numpy().tobytes() == oracle.encode(imgs[k], q, 1), k
    # the noise frames do not fit 12 000 bytes, the structured ones do: size 0, and the next frame follows at once
    tables, quant = sj.make_tables(quality=95.0)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    _, full = engine.encode_frames(frames, tables, header, sj.YUV_420)
    full = sorted(int(v) for v in full.cpu().numpy())
    stride = ((full[3] + full[4]) // 2) & ~15          # between the four structured frames and the three of noise
    assert full[3] + 64 < stride < full[4] - 64
    flat = torch.zeros(nf * stride, dtype=torch.uint8, device="cuda")
    sizes2 = torch.zeros(nf, dtype=torch.int64, device="cuda")
    offs2 = torch.zeros(nf + 1, dtype=torch.int64, device="cuda")
    engine.encode_frames_packed(frames, tables, header, sj.YUV_420, flat, sizes2, offs2, stride)
    torch.cuda.synchronize()
    sz, o = sizes2.cpu().numpy(), offs2.cpu().numpy()
    assert (sz == 0).any() and (sz > 0).any()
    for k in range(nf):
        assert int(o[k + 1] - o[k]) == ((int(sz[k]) + 15) & ~15)
        if sz[k]:
            assert flat[int(o[k]):int(o[k]) + int(sz[k])].cpu().numpy().tobytes() == oracle.encode(imgs[k], 95.0, 1), k


def test_exchange_of_packed_output_and_the_overflow_flag(engine, oracle):
    """One rank: the root codes PACKED output straight into the buffer the exchange gathers into -- gather_bytes has
    nothing to copy (d_packed == d_gathered + its offset) --; and a d_packed that was too small for
    sjpeg_hip_compact_streams shows as SJPEG_HIP_PACKED_OVERFLOW in d_offsets[nframes], which the rows carry to
    every rank: refused before anything is sent (the old check could not see it)."""
    w, h, nf = 200, 120, 5
    tables, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    imgs = [synth.g_struct(w, h, 900 + k) for k in range(nf)]
    frames = torch.from_numpy(np.stack(imgs)).cuda()
    stride = sj.frame_bound(w, h, sj.YUV_420, len(header))
    flat = torch.zeros(nf * stride, dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(nf, dtype=torch.int64, device="cuda")
    offs = torch.zeros(nf + 1, dtype=torch.int64, device="cuda")
    engine.encode_frames_packed(frames, tables, header, sj.YUV_420, flat, sizes, offs, stride)
    comm = sj.Comm(sj.comm_unique_id(), 0, 1)
    try:
        rows_dev = torch.zeros(2 * (nf + 2), dtype=torch.int64, device="cuda")
        rows, ro = comm.gather_rows(offs, sizes, nf, nf, rows_dev)
        before = flat.clone()
        comm.gather_bytes(0, flat, nf, rows, ro, flat)                    # in place: nothing moves
        torch.cuda.synchronize()
        assert torch.equal(before, flat)
        host, o = flat.cpu().numpy(), 0
        for k in range(nf):
            n = int(rows[0][2 + k])
            assert host[o:o + n].tobytes() == oracle.encode(imgs[k], 75.0, 1), k
            o += (n + 15) & ~15
        # compact_streams into a buffer that is one frame short
        out, sizes_s = engine.encode_frames(frames, tables, header, sj.YUV_420)
        need = int(offs[nf])
        small = torch.zeros(need - 16, dtype=torch.uint8, device="cuda")
        _, offs_s = sj.compact_streams(out, sizes_s, packed=small)
        torch.cuda.synchronize()
        assert int(offs_s[nf]) & ((1 << 63) - 1) == need and int(offs_s[nf]) < 0        # (int64 view of bit 63)
        with pytest.raises(sj.SjpegError):
            comm.gather_rows(offs_s, sizes_s, nf, nf, rows_dev)
    finally:
        comm.close()


def _sos_end(jpeg: bytes) -> int:
    """Offset of the first entropy-coded byte: behind the SOS segment (marker walk from the SOI)."""
    assert jpeg[:2] == b"\xff\xd8"
    at = 2
    while True:
        assert jpeg[at] == 0xFF, at
        marker = jpeg[at + 1]
        seg = (jpeg[at + 2] << 8) | jpeg[at + 3]
        at += 2 + seg
        if marker == 0xDA:
            return at


def test_headerless_segment_is_the_binding_of_integration_md(engine, golden_small, digests):
    """the reference keeps WriteSOS / WriteEOI (src/headers.cc:242-268) and replaces
    SinglePassScan() (src/enc.cc:437-443) by sjpeg_hip_encode_scan(header = NULL, 0, append_eoi = 0).  What that
    call returns must be exactly the reference file's bytes between the end of the SOS segment and the EOI marker
    -- stuffed, padded with 1-bits --, and the reference-written header + that segment + FF D9 the reference file.
    Reference bytes: tests/golden/small.npz (every geometry incl. the clipped ones, three colour modes)."""
    n = 0
    for key, want in golden_small.items():
        img, mode, q, method = golden_input(key)
        if method != 0:
            continue
        h, w = img.shape[:2]
        t, quant = sj.make_tables(quality=q)
        cut = _sos_end(want)
        assert want[-2:] == b"\xff\xd9"
        out, sizes = engine.encode_frames(dev(img), t, None, mode, append_eoi=False)
        torch.cuda.synchronize()
        seg = bytes(out[0, :int(sizes[0])].cpu().numpy())
        assert seg == want[cut:-2], key
        assert want[:cut] + seg + b"\xff\xd9" == want
        # the two halves one at a time: header without EOI, EOI without header
        out, sizes = engine.encode_frames(dev(img), t, want[:cut], mode, append_eoi=False)
        torch.cuda.synchronize()
        assert bytes(out[0, :int(sizes[0])].cpu().numpy()) == want[:-2], key
        out, sizes = engine.encode_frames(dev(img), t, None, mode, append_eoi=True)
        torch.cuda.synchronize()
        assert bytes(out[0, :int(sizes[0])].cpu().numpy()) == want[cut:], key
        n += 1
    assert n >= 180
    # a batch at full size, 1080p: clipped last MCU row, frames 0 and 1 of config #4
    frames = np.stack([synth.g_struct(1920, 1080, 7654321 + k) for k in range(2)])
    t, quant = sj.make_tables(quality=75.0)
    header = sj.make_header(1920, 1080, 1, quant)
    out, sizes = engine.encode_frames(torch.from_numpy(frames).cuda(), t, None, 1, append_eoi=False)
    torch.cuda.synchronize()
    for k in range(2):
        seg = bytes(out[k, :int(sizes[k])].cpu().numpy())
        whole = header + seg + b"\xff\xd9"
        assert hashlib.md5(whole).hexdigest() == digests[f"struct1080p_k{k}|420|q75|m0"]["md5"]


def _local_world(world, body):
    """Runs body(rank, comm) on `world` threads, every one with its own stream and a communicator of the LOCAL
    transport (sjpeg_hip_comm_create_local: the ranks are threads of this process, on this one device).
    Returns the per-rank results; an exception of any rank is re-raised here."""
    ident = os.urandom(128)
    results, errors = [None] * world, [None] * world

    def run(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                comm = sj.Comm(ident, r, world, local=True)
                try:
                    results[r] = body(r, comm)
                    torch.cuda.current_stream().synchronize()
                finally:
                    comm.close()
        except BaseException as e:          # noqa: BLE001 -- handed to the main thread
            errors[r] = e

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive(), "a rank of the local world hangs"
    return results, errors


@pytest.mark.parametrize("world,nframes,root", [(2, 7, 0), (3, 7, 0), (3, 8, 2), (2, 5, 1), (3, 2, 1), (3, 1, 0)])
def test_exchange_c_abi_with_several_ranks_on_one_device(oracle, world, nframes, root):
    """sjpeg_hip_gather_rows / _bytes / _streams with MORE THAN ONE rank (VERDICT r04 missing #1): their N > 1
    code -- per-rank receive offsets, grouped receives of exact lengths, the root that codes in place, a root that
    is not rank 0, ragged ranks, a rank with no frame at all -- on the local transport, the ranks being threads
    with a stream and an engine each.  Frame k is coded by rank k % world (sjpeg_amd.dist.shard_frames); the
    gathered buffer must be, byte for byte, what the gloo twin of the protocol (sjpeg_amd/dist.py, checked by
    tests/test_dist_cpu.py) holds: every rank's packed block -- its frames at multiples of 16, zero padding --
    in rank order, without gaps.  Reference: none (no communication backend)."""
    from sjpeg_amd.dist import shard_frames
    from test_dist_cpu import _compact_torch
    w, h, q = 200, 120, 75.0
    imgs = [synth.g_struct(w + 0, h, 4000 + k) if k % 3 else synth.g_noise(w, h, 4000 + k) for k in range(nframes)]
    want_frames = [oracle.encode(im, q, 1) for im in imgs]
    per_max = (nframes + world - 1) // world
    tables, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    stride = sj.frame_bound(w, h, sj.YUV_420, len(header))
    # the twin's buffer, from the oracle's bytes
    blocks = []
    for r in range(world):
        ids = shard_frames(nframes, r, world)
        if not ids:
            blocks.append(b"")
            continue
        out = torch.zeros((len(ids), stride), dtype=torch.uint8)
        sz = torch.tensor([len(want_frames[k]) for k in ids], dtype=torch.int64)
        for i, k in enumerate(ids):
            out[i, :len(want_frames[k])] = torch.from_numpy(np.frombuffer(want_frames[k], np.uint8).copy())
        need = int(((sz + 15) & ~15).sum())
        blocks.append(_compact_torch(out, sz, len(ids), need)[0].numpy().tobytes())
    want = b"".join(blocks)

    def body(rank, comm, in_place, one_call):
        ids = shard_frames(nframes, rank, world)
        n_local = len(ids)
        eng = sj.Engine(0)
        # the root that is rank 0 may code straight into the buffer the others are received behind
        cap = len(want) + 64
        room = max(n_local, 1) * stride + (cap if (in_place and rank == root) else 0)
        flat = torch.full((room,), 0x5A, dtype=torch.uint8, device="cuda")
        sizes = torch.zeros(max(n_local, 1), dtype=torch.int64, device="cuda")
        offs = torch.zeros(per_max + 1, dtype=torch.int64, device="cuda")
        if n_local:
            frames = torch.from_numpy(np.stack([imgs[k] for k in ids])).cuda()
            eng.encode_frames_packed(frames, tables, header, sj.YUV_420, flat, sizes, offs[:n_local + 1], stride)
        rows_dev = torch.zeros((world + 1) * (per_max + 2), dtype=torch.int64, device="cuda")
        if one_call:
            gathered = torch.full((cap,), 0xC3, dtype=torch.uint8, device="cuda") if rank == root else cap
            rows, ro = comm.gather_streams(root, flat, offs, sizes, n_local, per_max, rows_dev, gathered)
        else:
            rows, ro = comm.gather_rows(offs, sizes, n_local, per_max, rows_dev)
            assert int(ro[world]) == len(want)
            gathered = None
            if rank == root:
                gathered = flat if in_place else torch.full((int(ro[world]),), 0xC3, dtype=torch.uint8, device="cuda")
            comm.gather_bytes(root, flat, per_max, rows, ro, gathered)
        torch.cuda.current_stream().synchronize()
        # every rank has the same rows: sizes of every frame of every rank
        for r in range(world):
            rids = shard_frames(nframes, r, world)
            assert int(rows[r][1]) == len(rids) and [int(v) for v in rows[r][2:2 + len(rids)]] == [len(want_frames[k]) for k in rids]
            assert int(ro[r]) == sum(len(b) for b in blocks[:r])
        if rank != root:
            return None
        return gathered[:len(want)].cpu().numpy().tobytes()

    for in_place, one_call in ((False, False), (True, False), (False, True)):
        if in_place and root != 0:
            continue                                  # (only a root whose block comes first can code in place)
        results, errors = _local_world(world, lambda r, c: body(r, c, in_place, one_call))
        assert errors == [None] * world, errors
        assert results[root] == want, (in_place, one_call)
        assert all(results[r] is None for r in range(world) if r != root)
    # the frames, out of the gathered buffer, against the oracle (what a consumer does with rows + offsets)
    o = 0
    for r in range(world):
        for k in shard_frames(nframes, r, world):
            assert want[o:o + len(want_frames[k])] == want_frames[k]
            o += (len(want_frames[k]) + 15) & ~15


def test_exchange_refusals_reach_every_rank_of_a_local_world(oracle):
    """Three ranks: (a) one frame of rank 1 does not fit its output slot (size 0) -- sjpeg_hip_gather_rows returns
    SJPEG_HIP_ECAPACITY on EVERY rank, nothing is sent, nobody waits; (b) the root's capacity is too small for the
    total -- sjpeg_hip_gather_streams refuses on every rank before anybody sends; (c) the same communicators then
    do a good exchange: a refusal leaves no message behind."""
    from sjpeg_amd.dist import shard_frames
    world, nframes, root = 3, 6, 0
    w, h, q = 200, 120, 75.0
    imgs = [synth.g_struct(w, h, 5000 + k) for k in range(nframes)]
    imgs[4] = synth.g_noise(w, h, 5004)                                     # rank 1's second frame: three times the bytes
    want_frames = [oracle.encode(im, q, 1) for im in imgs]
    tables, quant = sj.make_tables(quality=q)
    header = sj.make_header(w, h, sj.YUV_420, quant)
    per_max = 2
    tight = (max(len(want_frames[k]) for k in range(nframes) if k != 4) + 64 + 15) & ~15
    assert tight < len(want_frames[4])
    full = sj.frame_bound(w, h, sj.YUV_420, len(header))
    total = sum((len(f) + 15) & ~15 for f in want_frames)

    def body(rank, comm):
        ids = shard_frames(nframes, rank, world)
        eng = sj.Engine(0)
        frames = torch.from_numpy(np.stack([imgs[k] for k in ids])).cuda()
        log = []
        for stride, cap in ((tight, total), (full, total - 16), (full, total)):
            flat = torch.zeros(len(ids) * stride, dtype=torch.uint8, device="cuda")
            sizes = torch.zeros(len(ids), dtype=torch.int64, device="cuda")
            offs = torch.zeros(per_max + 1, dtype=torch.int64, device="cuda")
            eng.encode_frames_packed(frames, tables, header, sj.YUV_420, flat, sizes, offs, stride)
            rows_dev = torch.zeros((world + 1) * (per_max + 2), dtype=torch.int64, device="cuda")
            gathered = torch.zeros(cap, dtype=torch.uint8, device="cuda") if rank == root else cap
            try:
                comm.gather_streams(root, flat, offs, sizes, len(ids), per_max, rows_dev, gathered)
                torch.cuda.current_stream().synchronize()
                log.append(gathered.cpu().numpy().tobytes() if rank == root else "sent")
            except sj.SjpegError as e:
                log.append("refused: " + ("size 0" if "size 0" in str(e) else "capacity" if "capacity" in str(e) else str(e)))
        return log

    results, errors = _local_world(world, body)
    assert errors == [None] * world, errors
    for r in range(world):
        assert results[r][0] == "refused: size 0" and results[r][1] == "refused: capacity", results[r][:2]
    got, o = results[root][2], 0
    for r in range(world):
        for k in shard_frames(nframes, r, world):
            assert got[o:o + len(want_frames[k])] == want_frames[k], k
            o += (len(want_frames[k]) + 15) & ~15
    assert o == total and results[1][2] == "sent" and results[2][2] == "sent"
