"""CPU-side tests of the product library: it loads, exports every symbol the headers declare,
its host-side preparation (quantizers, codes, headers, JPEG tools) equals the oracle's /
golden vectors, and without a GPU the encode path FAILS (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sjpeg_amd as sj

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_declared_symbols():
    lib = sj.lib()
    assert lib.SjpegVersion() == 0x000101
    assert lib.sjpeg_hip_abi_version() == 18
    declared = set()
    for hdr in ("include/sjpeg_hip.h", "include/sjpeg.h"):
        text = open(os.path.join(ROOT, hdr)).read()
        text = text.split("namespace sjpeg")[0]                   # C part only
        declared |= set(re.findall(r"\b(sjpeg_hip_[a-z_0-9]+|Sjpeg[A-Za-z]+)\s*\(", text))
    declared -= {"SjpegYUVMode"}
    assert declared, "header parse failed"
    assert declared == set(sj.EXPORTED_C_SYMBOLS), declared ^ set(sj.EXPORTED_C_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name


def test_documents_quote_the_header_abi_version():
    """INTEGRATION.md / DESIGN.md name the ABI version: it must be the header's (one was stale)."""
    ver = int(re.search(r"#define SJPEG_HIP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include/sjpeg_hip.h")).read()).group(1))
    assert ver == sj.lib().sjpeg_hip_abi_version()
    for doc in ("INTEGRATION.md", "DESIGN.md"):
        quoted = [int(v) for v in re.findall(r"ABI(?: version)? (\d+)", open(os.path.join(ROOT, doc)).read())]
        assert quoted and all(v == ver for v in quoted), (doc, quoted, ver)


def test_cxx_api_symbols_present():
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", sj.LIB_PATH]).decode()
    for sym in ("sjpeg::Encode(unsigned char const*, int, int, int, sjpeg::EncoderParam const&, sjpeg::ByteSink*)",
                "sjpeg::Encode(unsigned char const*, int, int, int, sjpeg::EncoderParam const&, unsigned char**)",
                "sjpeg::EncoderParam::EncoderParam(float)", "sjpeg::EncoderParam::SetQuantization",
                "sjpeg::EncoderParam::SetLimitQuantization", "sjpeg::MakeByteSink",
                "sjpeg::SearchHook::Update(float)"):
        assert sym in out, sym


@pytest.mark.parametrize("q", [0, 1, 10, 49.5, 50, 75, 90, 93, 99, 100])
def test_quantizer_tables_match_oracle(oracle, q):
    t, quant = sj.make_tables(quality=q)
    want = oracle.quality_matrices(q)
    assert (quant == want).all()
    for c in range(2):
        fq = oracle.finalize_quant(want[c])
        assert list(t.iquant[c]) == list(fq.iquant)
        assert list(t.bias[c]) == list(fq.bias)


def test_min_quant_and_bias(oracle):
    rng = np.random.RandomState(3)
    m = rng.randint(1, 256, (2, 64)).astype(np.uint8)
    mq = rng.randint(1, 64, (2, 64)).astype(np.uint8)
    t, quant = sj.make_tables(quant=m, min_quant=mq, q_bias=0x33)
    for c in range(2):
        fq = oracle.finalize_quant(m[c], mq[c], 0x33)
        assert list(quant[c]) == list(fq.quant)
        assert list(t.iquant[c]) == list(fq.iquant) and list(t.bias[c]) == list(fq.bias)


def test_default_huffman_codes_match_oracle(oracle):
    t, _ = sj.make_tables(quality=75)
    dc, ac = oracle.default_codes()
    assert (np.array(t.dc_codes) == dc).all()
    assert (np.array(t.ac_codes) == ac).all()


@pytest.mark.parametrize("mode", [1, 3, 4])
@pytest.mark.parametrize("dims", [(1, 1), (3840, 2160), (65535, 65535), (17, 13)])
def test_header_bytes_match_oracle(oracle, mode, dims):
    _, quant = sj.make_tables(quality=83)
    assert sj.make_header(dims[0], dims[1], mode, quant) == oracle.headers(dims[0], dims[1], mode, quant)


def test_header_is_prefix_of_golden(golden_small):
    key = "test128|128x128|420|q75|m0"
    _, quant = sj.make_tables(quality=75)
    h = sj.make_header(128, 128, 1, quant)
    assert golden_small[key][:len(h)] == h


def test_frame_bound():
    assert sj.frame_bound(0, 10, 1, 0) == 0
    assert sj.frame_bound(10, 10, 2, 0) == 0          # SHARP is not a scan mode
    b = sj.frame_bound(3840, 2160, 1, 619)
    assert b >= 32400 * 2560                          # reference bound per MCU (enc.cc:206-209)


def test_jpeg_tools_on_golden(golden_small):
    lib = sj.lib()
    for key in ("test128|128x128|420|q75|m0", "struct|17x13|444|q95|m0", "noise|140x99|400|q10|m0"):
        data = golden_small[key]
        _, dims, mname, q, _ = key.split("|")
        w0, h0 = (int(v) for v in dims.split("x"))
        w, h, is420 = C.c_int(), C.c_int(), C.c_int()
        assert lib.SjpegDimensions(data, len(data), C.byref(w), C.byref(h), C.byref(is420))
        assert (w.value, h.value, is420.value) == (w0, h0, int(mname == "420"))
        quant = np.zeros((2, 64), np.uint8)
        n = lib.SjpegFindQuantizer(data, len(data), quant.ctypes.data)
        assert n == (1 if mname == "400" else 2)
        _, want = sj.make_tables(quality=float(q[1:]))
        assert (quant[0] == want[0]).all()
        if n == 2:
            assert (quant[1] == want[1]).all()
        # every truncation is safe and never reports success on nonsense (unit_test.cc:456-484)
        for cut in range(0, min(len(data), 700), 7):
            lib.SjpegDimensions(data[:cut], cut, C.byref(w), C.byref(h), None)
            lib.SjpegFindQuantizer(data[:cut], cut, quant.ctypes.data)


def test_quant_matrix_and_quality_estimate(oracle):
    lib = sj.lib()
    for q in (1, 25, 50, 75, 90, 100):
        for chroma in (False, True):
            m = np.zeros(64, np.uint8)
            lib.SjpegQuantMatrix(float(q), chroma, m.ctypes.data)
            assert (m == oracle.quality_matrices(q)[int(chroma)]).all()
            est = lib.SjpegEstimateQuality(m.ctypes.data, chroma)
            assert abs(est - q) <= 1 or q in (1, 100)     # unit_test.cc:625-646


def test_live_jpeg_tools_vs_reference(reference, golden_small):
    lib = sj.lib()
    rng = np.random.RandomState(5)
    for _ in range(30):
        m = rng.randint(1, 256, 64).astype(np.uint8)
        for chroma in (False, True):
            a = lib.SjpegEstimateQuality(m.ctypes.data, chroma)
            b = reference.lib.ref_estimate_quality(m.ctypes.data, int(chroma))
            assert a == b


def test_adapt_quant_matches_oracle(oracle):
    """Host AnalyseHisto (product) vs the oracle's, on oracle-computed histograms."""
    from oracle import synth
    for (w, h, mode, q) in ((128, 96, 1, 75.0), (200, 120, 3, 90.0), (64, 64, 4, 50.0), (333, 211, 1, 30.0)):
        for gen in (synth.g_struct, synth.g_noise):
            img = gen(w, h, 5)
            hist = oracle.histogram(img, mode)
            _, quant = sj.make_tables(quality=q)
            want, qs = oracle.adapt_quant(hist, 1 if mode == 4 else 3, quant)
            t, got = sj.adapt_quant(hist, mode, quant)
            assert (got == want).all(), (w, h, mode, q)
            ncheck = 1 if mode == 4 else 2
            for c in range(ncheck):
                assert list(t.iquant[c]) == list(qs[c].iquant) and list(t.bias[c]) == list(qs[c].bias)


def test_optimal_huffman_matches_oracle(oracle):
    from oracle import synth
    rng = np.random.RandomState(4)
    cases = [oracle.symbol_stats(synth.g_struct(160, 96, 3), sj.make_tables(quality=75)[1], 0x78, 1),
             oracle.symbol_stats(synth.g_noise(96, 96, 3), sj.make_tables(quality=95)[1], 0x78, 3)]
    f = np.zeros((2, 272), np.uint32)
    f[:, :256] = rng.randint(0, 2, (2, 256)) * rng.randint(1, 1 << 20, (2, 256))   # sparse, huge spread
    f[:, 256:268] = rng.randint(1, 100, (2, 12))
    cases.append(f)
    g = np.zeros((2, 272), np.uint32)                   # Fibonacci counts: code lengths up to 27
    fib = [1, 1]                                        # bits, exercising the 16-bit limiter
    while len(fib) < 28:
        fib.append(fib[-1] + fib[-2])
    g[:, 1:29] = np.array(fib, np.uint32)
    g[:, 256:268] = np.array(fib[:12], np.uint32)
    cases.append(g)
    for freq in cases:
        t = sj.ScanTables()
        specs = sj.optimize_huffman(freq, 1, t)
        for tbl in range(2):
            for kind, off, size in ((0, 256, 12), (2, 0, 256)):
                bits, syms, n = oracle.build_optimal(freq[tbl, off:off + size], size)
                sp = specs[kind + tbl]
                assert sp.nsyms == n and list(sp.bits) == list(bits) and list(sp.syms[:n]) == list(syms)
                assert sum(sp.bits) == n and all(b >= 0 for b in sp.bits)


def test_optimal_huffman_ties_and_long_codes_match_oracle(oracle):
    # equal counts everywhere (the merge order is decided by the reference's tie rule, and the reserved
    # leaf is the lighter half of the first merge whatever its key), and power-of-two counts that make
    # codes of up to ~20 bits for the 16-bit limiter
    rng = np.random.RandomState(7)
    for it in range(60):
        f = np.zeros((2, 272), np.uint32)
        kind = it % 4
        if kind == 0:
            f[:, :256] = rng.randint(0, 3, (2, 256))
        elif kind == 1:
            f[:, :256] = rng.randint(0, 2, (2, 256)) * rng.randint(1, 5, (2, 256))
        elif kind == 2:
            f[:, :256] = (rng.randint(0, 4, (2, 256)) == 0) * (1 << rng.randint(0, 20, (2, 256)).astype(np.int64))
        else:
            f[:, :256] = rng.randint(0, 1 << 31, (2, 256)) * (rng.randint(0, 8, (2, 256)) == 0)
        f[:, 256:268] = rng.randint(0, 3, (2, 12))
        f[:, 256] = 1
        f[:, 0] = np.maximum(f[:, 0], 1)
        t = sj.ScanTables()
        specs = sj.optimize_huffman(f, 1, t)
        for tbl in range(2):
            for k, off, size in ((0, 256, 12), (2, 0, 256)):
                bits, syms, n = oracle.build_optimal(f[tbl, off:off + size], size)
                sp = specs[k + tbl]
                assert sp.nsyms == n and list(sp.bits) == [int(b) for b in bits]
                assert list(sp.syms[:n]) == [int(x) for x in syms][:n]


def test_optimal_huffman_survives_counts_that_break_the_depth_clamp():
    """Fibonacci-like counts over 40 symbols make a tree deeper than the 32 levels the builder keeps: the length
    histogram is no prefix code any more and K.2's limiter cannot repair it (the reference is undefined there).
    The table must still be a VALID one: every used symbol listed once, Kraft's sum at most 1 with the reserved
    all-ones code, no length over 16 (the early return left codes of more than 16 bits behind)."""
    f = np.zeros((2, 272), np.uint32)
    a, b = 1, 2
    for k in range(44):
        f[:, k] = min(a, (1 << 32) - 1)
        a, b = b, a + b
    f[:, 256] = 1
    t = sj.ScanTables()
    specs = sj.optimize_huffman(f, 1, t)
    for tbl in range(2):
        sp = specs[2 + tbl]
        bits = [int(x) for x in sp.bits]
        assert sp.nsyms == 44 and sum(bits) == 44
        assert sorted(sp.syms[:44]) == list(range(44))
        assert sum((c + (1 if l == max(i for i, v in enumerate(bits) if v) else 0)) * 2.0 ** -(l + 1) for l, c in enumerate(bits)) <= 1.0 + 1e-12


@pytest.mark.parametrize("nused", [255, 256, 100, 45])
def test_flat_fallback_with_every_symbol_used(nused):
    """The same breakdown with 255 / 256 used symbols (256 / 257 leaves with the reserved one): the fallback code
    must still fit the DHT's byte counts -- a complete code of two lengths, e.g. 255 of 8 bits + 1 of 9 -- and list
    every symbol once (one length for all leaves wrapped a uint8 to 0 / 1)."""
    f = np.zeros((2, 272), np.uint32)
    a, b = 1, 2
    for k in range(nused):
        f[:, k] = min(a, (1 << 32) - 1)
        a, b = b, a + b
    f[:, 256] = 1
    t = sj.ScanTables()
    specs = sj.optimize_huffman(f, 1, t)
    for tbl in range(2):
        sp = specs[2 + tbl]
        bits = [int(x) for x in sp.bits]
        assert sp.nsyms == nused and sum(bits) == nused, (sp.nsyms, bits)
        assert sorted(sp.syms[:nused]) == list(range(nused))
        longest = max(i for i, v in enumerate(bits) if v)
        kraft = sum((c + (1 if l == longest else 0)) * 2.0 ** -(l + 1) for l, c in enumerate(bits))
        assert kraft <= 1.0 + 1e-12, (bits, kraft)
        # the codes the table installs are distinct prefix-free words of the listed lengths
        codes = [(int(t.ac_codes[tbl][s]) >> 16, int(t.ac_codes[tbl][s]) & 0xff) for s in range(nused)]
        assert all(1 <= n <= 16 for _, n in codes)
        words = sorted(format(c, "0%db" % n) for c, n in codes)
        assert all(not words[i + 1].startswith(words[i]) for i in range(len(words) - 1))
        assert all("0" in w for w in words)                       # no all-ones code (T.81 C: reserved)


def test_shipped_riskiness_table_is_the_reference_table():
    """sjpeg_amd/csrc/riskiness.bin (reference DATA, shipped with attribution: riskiness.NOTICE) is the table
    of the reference build, where there is one; its digest is pinned either way."""
    import hashlib
    data = open(os.path.join(sj.CSRC, "riskiness.bin"), "rb").read()
    assert len(data) == 117649 and hashlib.md5(data).hexdigest() == "56618d0235dbfddab20358969d2dd241"
    from oracle import refso
    if refso.available():
        import ctypes
        ref = bytes((ctypes.c_uint8 * 117649).in_dll(ctypes.CDLL(refso.REF_SO), "_ZN5sjpeg15kSharpnessScoreE"))
        assert ref == data


def test_header_with_custom_tables_matches_golden(golden_small, oracle, img128):
    # method-1 golden stream: its header must be reproduced from oracle statistics
    want = golden_small["test128|128x128|420|q75|m1"]
    _, quant = sj.make_tables(quality=75)
    freq = oracle.symbol_stats(img128, quant, 0x78, 1)
    t, _ = sj.make_tables(quality=75)
    specs = sj.optimize_huffman(freq, 1, t)
    h = sj.make_header_ex(128, 128, 1, quant, specs)
    assert want[:len(h)] == h


def test_invalid_arguments_fail_cleanly():
    # reference: unit_test.cc:165-193 -- all of these return 0 without touching the GPU
    img = np.zeros((8, 8, 3), np.uint8)
    lib = sj.lib()
    out = C.POINTER(C.c_uint8)()
    assert lib.SjpegEncode(None, 8, 8, 24, C.byref(out), 75.0, 0, 1) == 0
    assert lib.SjpegEncode(img.ctypes.data, 0, 8, 24, C.byref(out), 75.0, 0, 1) == 0
    assert lib.SjpegEncode(img.ctypes.data, 8, -1, 24, C.byref(out), 75.0, 0, 1) == 0
    assert lib.SjpegEncode(img.ctypes.data, 8, 8, 23, C.byref(out), 75.0, 0, 1) == 0
    assert lib.SjpegEncode(img.ctypes.data, 8, 8, -23, C.byref(out), 75.0, 0, 1) == 0
    assert lib.SjpegEncode(img.ctypes.data, 8, 8, 24, None, 75.0, 0, 1) == 0


def test_no_gpu_means_failure_not_fallback():
    if sj.device_count() > 0:
        pytest.skip("a GPU is present")
    img = np.zeros((16, 16, 3), np.uint8)
    assert sj.SjpegEncode(img, 75, 0, sj.YUV_420) is None
    assert "no HIP device" in sj.last_error()
    with pytest.raises(sj.SjpegError):
        sj.Engine(0)


def test_product_never_references_oracle():
    # the shipped library and binding must not import, link or load anything under oracle/
    for rel in ("sjpeg_amd/__init__.py", "sjpeg_amd/dist.py"):
        assert "oracle" not in open(os.path.join(ROOT, rel)).read().replace("no pure-Python", "")
    for f in os.listdir(os.path.join(ROOT, "sjpeg_amd", "csrc")):
        if f.endswith((".cc", ".hip", ".h", "Makefile")):
            text = open(os.path.join(ROOT, "sjpeg_amd", "csrc", f)).read()
            assert "oracle/" not in text and "sjpeg_oracle" not in text, f
    import subprocess
    needed = subprocess.check_output(["readelf", "-d", sj.LIB_PATH]).decode()
    assert "oracle" not in needed and "sjpeg_ref" not in needed


def test_metadata_headers_match_reference_digests():
    """APP markers, EXIF, chunked ICC, XMP and extended XMP (MD5-tied extension chunks,
    reference src/headers.cc:63-180): header with metadata + the oracle's entropy segment must be
    the reference's file (tests/golden/extra.json, generated from the real reference)."""
    import importlib.util
    import json
    from oracle import orc, synth
    here = os.path.join(ROOT, "tests", "golden")
    spec = importlib.util.spec_from_file_location("make_golden_extra", os.path.join(here, "make_golden_extra.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    dig = json.load(open(os.path.join(here, "extra.json")))
    img = synth.g_struct(40, 24, 3)
    quant = sj.make_tables(quality=80.0)[1]
    plain = orc.oracle().encode(img, 80.0, 1)
    plain_hdr = sj.make_header(40, 24, 1, quant)
    assert plain.startswith(plain_hdr)
    n = 0
    for key, kw in mod.meta_cases():
        hdr = sj.make_header_meta(40, 24, 1, quant, **kw)
        assert hdr is not None, key
        full = hdr + plain[len(plain_hdr):]
        assert len(full) == dig[key]["size"] and synth.md5(full) == dig[key]["md5"], key
        n += 1
    assert n == 6
    # invalid metadata fails like the reference (src/headers.cc:77,95,122-125)
    assert sj.make_header_meta(40, 24, 1, quant, exif=b"e" * 70000) is None
    assert sj.make_header_meta(40, 24, 1, quant, iccp=b"i" * (256 * 65519)) is None
    assert sj.make_header_meta(40, 24, 1, quant, xmp=b"x" * 70000) is None          # no HasExtendedXMP
