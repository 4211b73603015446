"""Generates tests/golden/extra.json from the REAL reference (oracle/_ref/libsjpeg_ref.so): size + MD5
of its output for the other input layouts of the API (EncodeBGRA/RGBA/Gray/YUV444/YUV420/NV12/NV21)
and for the multi-pass size / PSNR search.  Dev container only:

    make -C oracle ref && python tests/golden/make_golden_extra.py

Inputs are rebuilt from seeds by `source_planes()` / `search_cases()` below, which the tests import.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_SIZES = [(1, 1), (17, 13), (40, 9), (97, 61)]
SRC_SETTINGS = [(75.0, False, False), (40.0, True, True), (92.0, False, True), (60.0, True, False)]


def source_planes(fmt, w, h):
    """Deterministic planes of layout `fmt` (oracle ORC_SRC_* numbering, 1..7)."""
    rng = np.random.RandomState(1000 * fmt + 7 * w + h)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    shapes = {1: [(h, 4 * w)], 2: [(h, 4 * w)], 3: [(h, w)], 4: [(h, w)] * 3,
              5: [(h, w), (ch, cw), (ch, cw)], 6: [(h, w), (ch, 2 * cw)], 7: [(h, w), (ch, 2 * cw)]}[fmt]
    planes = [rng.randint(0, 256, s).astype(np.uint8) for s in shapes]
    if w >= 40:            # smoother content so that runs and large coefficients both occur
        planes = [(p // 4 + np.arange(p.shape[1])[None, :] // 3).astype(np.uint8) for p in planes]
    return planes


def source_cases():
    for fmt in range(1, 8):
        for (w, h) in SRC_SIZES:
            for mode in ((1, 3, 4) if fmt in (1, 2) else (1,)):
                for (q, huff, adapt) in SRC_SETTINGS:
                    yield f"src{fmt}|{w}x{h}|mode{mode}|q{q:g}|h{int(huff)}|a{int(adapt)}", fmt, w, h, mode, q, huff, adapt


def search_cases():
    """(key, picture, quality seed, mode, huffman, adaptive, target_mode, target, passes, tolerance)"""
    pics = {"struct141x99": synth.g_struct(141, 99, 4242), "noise64x48": synth.g_noise(64, 48, 99)}
    for pname, img in pics.items():
        for mode in (1, 3, 4):
            for huff in (False, True):
                for adapt in (False, True):
                    for tm, targets, tol in ((1, (1500.0, 4000.0, 9000.0), 1.0), (2, (30.0, 38.0, 45.0), 0.1)):
                        for target in targets:
                            for passes in (2, 6, 10):
                                key = f"search|{pname}|mode{mode}|h{int(huff)}|a{int(adapt)}|t{tm}|{target:g}|p{passes}"
                                yield key, img, 60.0, mode, huff, adapt, tm, target, passes, tol


def xmp_long(n):
    """A long XMP packet with the HasExtendedXMP attribute the extension mechanism needs."""
    head = b'<x:xmpmeta xmlns:x="adobe:ns:meta/"><rdf:Description xmpNote:HasExtendedXMP="' + b"0" * 32 + b'"/>'
    return head + bytes(((i * 7 + 3) & 0x7f) | 0x20 for i in range(n - len(head)))


def meta_cases():
    """(key, kwargs of metadata) on the 40x24 G_struct picture, q80 4:2:0 method 0"""
    yield "meta|none", dict()
    yield "meta|exif+icc2+xmp+app", dict(exif=b"II*\0abc" * 10, iccp=b"i" * 70000, xmp=b"<x/>",
                                         app_markers=b"\xff\xe5\x00\x04zz")
    yield "meta|xmp70000", dict(xmp=xmp_long(70000))
    yield "meta|xmp200000", dict(xmp=xmp_long(200000))
    yield "meta|xmp66000|split200", dict(xmp=xmp_long(66000), xmp_split_point=200)
    yield "meta|xmp_exact_two_chunks", dict(xmp=xmp_long(65503 + 65458))


def main():
    from oracle import refso
    r = refso.ref()
    dig = {}
    for key, fmt, w, h, mode, q, huff, adapt in source_cases():
        out = r.encode_src(fmt, source_planes(fmt, w, h), w, h, q, mode, huff, adapt)
        dig[key] = dict(size=len(out), md5=synth.md5(out))
    for key, img, q, mode, huff, adapt, tm, target, passes, tol in search_cases():
        out = r.encode_search(img, q, mode, huff, adapt, tm, target, passes, tol)
        dig[key] = dict(size=len(out), md5=synth.md5(out))
    img = synth.g_struct(40, 24, 3)
    for key, kw in meta_cases():
        out = r.encode_meta(img, 80.0, 1, **kw)
        dig[key] = dict(size=len(out), md5=synth.md5(out))
    with open(os.path.join(HERE, "extra.json"), "w") as f:
        json.dump(dig, f, indent=0, sort_keys=True)
    print("wrote", len(dig), "digests")


if __name__ == "__main__":
    main()
