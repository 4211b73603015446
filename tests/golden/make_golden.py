"""Generates tests/golden/* from the REAL reference (oracle/_ref/libsjpeg_ref.so, built from
/root/reference by oracle/Makefile).  Run in the dev container only:

    make -C oracle ref && python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs, no reference code):
  test128.rgb        128x128 packed RGB decoded (PIL) from the reference's own test image
                     tests/testdata/test_exif_xmp.png  (BASELINE.json config #1 input)
  small.npz          expected JPEG bytes of the reference for small inputs
                     (keys "<name>|<w>x<h>|<mode>|q<quality>|m<method>")
  digests.json       size + MD5 of the reference output for the full-size BASELINE configs,
                     plus the quantization matrices of the recompress recipe (config #5)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refso, synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_TESTDATA = "/root/reference/tests/testdata"

SMALL_SIZES = [(1, 1), (7, 5), (8, 8), (16, 16), (17, 13), (33, 21), (40, 9), (15, 40), (64, 24),
               (140, 99)]
SMALL_Q = [10.0, 75.0, 95.0]
MODES = {"420": 1, "444": 3, "400": 4}


def main():
    r = refso.ref()
    from PIL import Image
    img128 = np.asarray(Image.open(os.path.join(REF_TESTDATA, "test_exif_xmp.png")).convert("RGB"))
    assert img128.shape == (128, 128, 3)
    img128.tofile(os.path.join(HERE, "test128.rgb"))

    small = {}
    for name, gen in (("struct", synth.g_struct), ("noise", synth.g_noise)):
        for (w, h) in SMALL_SIZES:
            img = gen(w, h, 7654321 + w)
            for mname, mode in MODES.items():
                for q in SMALL_Q:
                    out = r.encode(img, q, 0, mode)
                    small[f"{name}|{w}x{h}|{mname}|q{q:g}|m0"] = np.frombuffer(out, np.uint8)
    for mname, mode in MODES.items():
        for method in (0, 1, 3, 4, 7):
            out = r.encode(img128, 75.0, method, mode)
            small[f"test128|128x128|{mname}|q75|m{method}"] = np.frombuffer(out, np.uint8)
    # bottom-up input (negative stride) must equal the flipped picture (unit_test.cc:311-342)
    np.savez_compressed(os.path.join(HERE, "small.npz"), **small)

    dig = {}

    def put(key, data, **extra):
        dig[key] = dict(size=len(data), md5=synth.md5(data), **extra)

    put("test128|compress_q75", r.compress(img128, 75.0))
    s4k = synth.g_struct(3840, 2160)
    n4k = synth.g_noise(3840, 2160)
    dig["input|struct4k"] = dict(md5=synth.md5(s4k))
    dig["input|noise4k"] = dict(md5=synth.md5(n4k))
    for mname, mode in MODES.items():
        put(f"struct4k|{mname}|q75|m0", r.encode(s4k, 75.0, 0, mode))
        put(f"noise4k|{mname}|q75|m0", r.encode(n4k, 75.0, 0, mode))
    for method in (1, 3, 4):
        put(f"struct4k|420|q75|m{method}", r.encode(s4k, 75.0, method, 1))
    # G_noise 4K with default parameters (method 4); 4:4:4 and 4:0:0 beside it, and the
    # structured picture in those two modes (wide levels, two bit windows per segment, 4:4:4 parts)
    for mname, mode in MODES.items():
        put(f"noise4k|{mname}|q75|m4", r.encode(n4k, 75.0, 4, mode))
        if mname != "420":
            put(f"struct4k|{mname}|q75|m4", r.encode(s4k, 75.0, 4, mode))
    # config #5: recompress recipe (examples/sjpeg.cc:262-286), method-0 variant
    src = r.encode_param(s4k, quality=92.0, yuv_mode=1, huffman=True, adaptive=True)
    nq, qm = r.find_quantizer(src)
    put("recompress|source_q92_default", src, nq=nq)
    rec0 = r.encode_param(s4k, quality=75.0, yuv_mode=1, huffman=False, adaptive=False, quant=qm,
                          reduction=90.0, limit_quant=True)
    put("recompress|r90|m0", rec0, source_quant=qm.reshape(-1).tolist())
    rec4 = r.encode_param(s4k, quality=75.0, yuv_mode=1, huffman=True, adaptive=True, quant=qm,
                          reduction=90.0, limit_quant=True)
    put("recompress|r90|default", rec4)
    del s4k, n4k
    # config #4: 64 x 1080p
    import hashlib
    cat = hashlib.md5()
    total = 0
    for k in range(64):
        f = synth.g_struct(1920, 1080, 7654321 + k)
        out = r.encode(f, 75.0, 0, 1)
        cat.update(out)
        total += len(out)
        if k in (0, 1, 63):
            put(f"struct1080p_k{k}|420|q75|m0", out)
    dig["struct1080p_k0..63_concat|420|q75|m0"] = dict(size=total, md5=cat.hexdigest())
    # config #3: 8K 4:4:4 q90
    s8k = synth.g_struct(7680, 4320)
    put("struct8k|444|q90|m0", r.encode(s8k, 90.0, 0, 3))
    # SURVEY 8c / BASELINE "C3'": the same frame with default parameters (method 4)
    put("struct8k|444|q90|m4", r.encode(s8k, 90.0, 4, 3))
    put("struct8k|444|q90|m1", r.encode(s8k, 90.0, 1, 3))
    put("struct8k|444|q90|m3", r.encode(s8k, 90.0, 3, 3))
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(dig, f, indent=1, sort_keys=True)
    print("wrote", len(small), "small vectors,", len(dig), "digests")


if __name__ == "__main__":
    main()
