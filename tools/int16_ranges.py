#!/usr/bin/env python3
"""Machine-checked ranges of K1's packed arithmetic (sjpeg_amd/csrc/scan_device.h).  Runs anywhere (no GPU).

The reference's scalar fDCT works in `int` (src/fdct.cc:161-209) and has nothing to prove; the device code
works on int16 PAIRS and on 24-bit multiplies and has: solid red / blue pictures were coded wrong for three
rounds because a sum of four +128 columns reached 32768 in an int16 lane.  This script replaces "the fuzz found
nothing" by "it cannot wrap":

1. PROOF.  Every intermediate of `fdct_col8_pk`, `fdct_row8_pk` and `row_quant` is written as an affine form
   of the block's 64 samples -- exact rational coefficients, plus an error interval for the floors of the
   fixed-point multiplies -- following the device code statement by statement.  The extreme of an affine form over
   a box is attained at a corner, so `sum(c_i > 0 ? c_i * hi : c_i * lo)` + error is a TIGHT bound (samples are
   independent: any corner is a picture).  Each value is checked against the type it lives in: an int16 lane,
   a 24-bit multiplier operand, a 32-bit product or accumulator, a u16 operand of v_mad_u32_u16.  Sample ranges:
   luma and planar sources -128 .. 127, chroma from RGB -127 .. 128 (pure blue / red reach +128, yellow / cyan
   -127; src/colors_rgb.cc:785-828) -- the proof is run for BOTH, not for their union (the union would overflow:
   128 * 256 = 32768).  The colour conversion (`luma_pair`, `cb_sum`, `cr_sum`) is linear in R, G, B: its sums
   are evaluated at the corners of the RGB cube.
2. MODEL.  The same statements with the hardware's wrap-around semantics (numpy int16 / int32) are run on the
   extremal sample patterns the proof found and on random lattices, against the oracle's `orc_fdct`
   (oracle/sjpeg_oracle.c, the reference's scalar statements): a wrong range table would show as a mismatch.
3. PATTERNS.  The corner patterns that drive the tightest lanes go to tests/golden/extremal_patterns.json; the GPU
   tests (tests/test_gpu_parity.py::test_extremal_patterns_*) paint them with blue|yellow, red|cyan and
   black|white pixels and compare coefficients and bytes with the oracle.

  python tools/int16_ranges.py [--write-patterns] [--table profiles/r04/int16_ranges.txt]
Exit code 1 if any lane can leave its type or the model differs from the oracle.
"""
import argparse
import json
import os
import sys
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

I16 = (-(1 << 15), (1 << 15) - 1)
I24 = (-(1 << 23), (1 << 23) - 1)
I32 = (-(1 << 31), (1 << 31) - 1)
U16 = (0, (1 << 16) - 1)
U32 = (0, (1 << 32) - 1)

# row tables of the reference (src/fdct.cc:28-35,599-606): C1..C7 per row index
ROW_TABLES = [
    (22725, 21407, 19266, 16384, 12873, 8867, 4520),
    (31521, 29692, 26722, 22725, 17855, 12299, 6270),
    (29692, 27969, 25172, 21407, 16819, 11585, 5906),
    (26722, 25172, 22654, 19266, 15137, 10426, 5315),
    (22725, 21407, 19266, 16384, 12873, 8867, 4520),
    (26722, 25172, 22654, 19266, 15137, 10426, 5315),
    (29692, 27969, 25172, 21407, 16819, 11585, 5906),
    (31521, 29692, 26722, 22725, 17855, 12299, 6270),
]


# ------------------------------------------------------------------------------------------------
# 1. affine forms

class Aff:
    """value = sum(c[i] * x[i]) + e, e in [elo, ehi]; x = the block's 64 samples (row-major)."""
    __slots__ = ("c", "elo", "ehi")

    def __init__(self, c, elo=Fraction(0), ehi=Fraction(0)):
        self.c, self.elo, self.ehi = c, Fraction(elo), Fraction(ehi)

    @staticmethod
    def var(i):
        c = [Fraction(0)] * 64
        c[i] = Fraction(1)
        return Aff(c)

    def __add__(self, o):
        return Aff([a + b for a, b in zip(self.c, o.c)], self.elo + o.elo, self.ehi + o.ehi)

    def __sub__(self, o):
        return Aff([a - b for a, b in zip(self.c, o.c)], self.elo - o.ehi, self.ehi - o.elo)

    def scale(self, k):
        k = Fraction(k)
        lo, hi = self.elo * k, self.ehi * k
        return Aff([a * k for a in self.c], min(lo, hi), max(lo, hi))

    def plus(self, k):
        return Aff(list(self.c), self.elo + k, self.ehi + k)

    def floor_shift(self, bits):
        """floor(value / 2^bits): value / 2^bits - frac, frac in [0, 1)"""
        s = self.scale(Fraction(1, 1 << bits))
        return Aff(s.c, s.elo - 1, s.ehi)

    def bounds(self, lo, hi):
        mx = sum(a * (hi if a > 0 else lo) for a in self.c) + self.ehi
        mn = sum(a * (lo if a > 0 else hi) for a in self.c) + self.elo
        return mn, mx

    def corner(self, want_max=True):
        """64-bit mask: bit i set = sample i at the upper end of its range (the corner that maximises the form;
        for want_max=False the one that minimises it)."""
        m = 0
        for i, a in enumerate(self.c):
            if (a > 0) == want_max and a != 0:
                m |= 1 << i
        return m


class Checker:
    def __init__(self, lo, hi, label):
        self.lo, self.hi, self.label = lo, hi, label
        self.rows = []          # (name, type, min, max, limit, margin)
        self.bad = []
        self.patterns = {}      # name -> (mask_max, mask_min)

    def check(self, name, aff, typ, tname, keep_pattern=False):
        mn, mx = aff.bounds(self.lo, self.hi)
        margin = min(mn - typ[0], typ[1] - mx)
        self.rows.append((name, tname, mn, mx, typ, margin))
        if margin < 0:
            self.bad.append((name, tname, mn, mx))
        if keep_pattern:
            self.patterns[name] = (aff.corner(True), aff.corner(False))
        return aff

    def check_const(self, name, k, typ, tname):
        if not (typ[0] <= k <= typ[1]):
            self.bad.append((name, tname, k, k))
        # (an unsigned value has no lower edge to fall off)
        self.rows.append((name, tname, Fraction(k), Fraction(k), typ, Fraction(typ[1] - k if typ[0] == 0 else min(k - typ[0], typ[1] - k))))


def pk_mulhi(ck, name, a, K):
    """pk_mulhi(a, K): two v_mul_i32_i24 + v_perm taking bits 16..31 -- floor(a * K / 65536) as int16."""
    ck.check(name + ": multiplier operand", a, I24, "i24")
    ck.check_const(name + ": constant", K, I24, "i24")
    ck.check(name + ": 32-bit product", a.scale(K), I32, "i32")
    return ck.check(name + ": result lane", a.scale(K).floor_shift(16), I16, "i16")


def col_pass(ck, x, tag):
    """fdct_col8_pk on one column x[0..7] (scan_device.h), statement by statement."""
    c = lambda n, v, keep=False: ck.check(f"col {tag} {n}", v, I16, "i16", keep)
    d07, s07 = c("d07", x[0] - x[7]), c("s07", x[0] + x[7])
    d25, s25 = c("d25", x[2] - x[5]), c("s25", x[2] + x[5])
    d34, s34 = c("d34", x[3] - x[4]), c("s34", x[3] + x[4])
    d16, s16 = c("d16", x[1] - x[6]), c("s16", x[1] + x[6])
    ed, es = c("ed", s07 - s34), c("es", s07 + s34)
    fd, fs = c("fd", s16 - s25), c("fs", s16 + s25)
    a, b = c("a=es<<3", es.scale(8)), c("b=fs<<3", fs.scale(8))
    r0 = c("r0", a + b, True)
    r4 = c("r4", a - b, True)
    ed, fd = c("ed<<3", ed.scale(8)), c("fd<<3", fd.scale(8))
    d34, d07 = c("d34<<3", d34.scale(8)), c("d07<<3", d07.scale(8))
    r2 = c("r2", pk_mulhi(ck, f"col {tag} mulhi(fd,27146)", fd, 27146) + ed, True)
    r6 = c("r6", pk_mulhi(ck, f"col {tag} mulhi(ed,27146)", ed, 27146) - fd, True)
    dm, dp = c("d16-d25", d16 - d25), c("d16+d25", d16 + d25)
    od = pk_mulhi(ck, f"col {tag} od", dm, 23170 << 4)
    os_ = pk_mulhi(ck, f"col {tag} os", dp, 23170 << 4)
    p3, p1 = c("p3", d34 - od), c("p1", d34 + od)
    p0, p2 = c("p0", d07 - os_), c("p2", d07 + os_)
    u3 = pk_mulhi(ck, f"col {tag} u3", p3, 65536 - 21746)
    t4 = pk_mulhi(ck, f"col {tag} t4", p0, 65536 - 21746)
    t5 = pk_mulhi(ck, f"col {tag} t5", p2, 13036)
    # ~p2 = -p2 - 1 (bitwise NOT of an int16 lane never leaves the lane)
    r1 = c("r1", pk_mulhi(ck, f"col {tag} mulhi(p1,13036)", p1, 13036) - (p2.scale(-1).plus(-1)), True)
    r3 = c("r3", p0 + (u3.scale(-1).plus(-1)), True)
    r5 = c("r5", p3 + t4, True)
    r7 = c("r7", t5 - p1, True)
    return [r0, r1, r2, r3, r4, r5, r6, r7]


def row_pass(ck, v, r):
    """fdct_row8_pk on row r (v[0..7] = the column pass' outputs of that row), then row_quant's range checks."""
    C1, C2, C3, C4, C5, C6, C7 = ROW_TABLES[r]
    c = lambda n, val, keep=False: ck.check(f"row {r} {n}", val, I16, "i16", keep)
    a0, a1 = c("a0=x0+x7", v[0] + v[7], True), c("a1=x1+x6", v[1] + v[6], True)
    b0, b1 = c("b0=x0-x7", v[0] - v[7], True), c("b1=x1-x6", v[1] - v[6], True)
    a3, a2 = c("a3=x3+x4", v[3] + v[4], True), c("a2=x2+x5", v[2] + v[5], True)
    b3, b2 = c("b3=x3-x4", v[3] - v[4], True), c("b2=x2-x5", v[2] - v[5], True)
    c1, c3 = c("c1=a0-a3", a0 - a3, True), c("c3=a1-a2", a1 - a2, True)
    acc = [None] * 8
    d = lambda n, val: ck.check(f"row {r} acc{n}", val, I32, "i32", True)
    # every partial sum of a v_dot2_i32_i16 chain is a 32-bit value too
    t = ck.check(f"row {r} acc0 first dot", (a0 + a1).scale(C4), I32, "i32")
    acc[0] = d(0, t + (a3 + a2).scale(C4))
    t = ck.check(f"row {r} acc4 first dot", (a0 - a1).scale(C4), I32, "i32")
    acc[4] = d(4, t + (a3 - a2).scale(C4))
    acc[2] = d(2, c1.scale(C2) + c3.scale(C6))
    acc[6] = d(6, c1.scale(C6) - c3.scale(C2))
    t = ck.check(f"row {r} acc1 first dot", b0.scale(C1) + b1.scale(C3), I32, "i32")
    acc[1] = d(1, t + b3.scale(C7) + b2.scale(C5))
    t = ck.check(f"row {r} acc3 first dot", b0.scale(C3) - b1.scale(C7), I32, "i32")
    acc[3] = d(3, t - b3.scale(C5) - b2.scale(C1))
    t = ck.check(f"row {r} acc5 first dot", b0.scale(C5) - b1.scale(C1), I32, "i32")
    acc[5] = d(5, t + b3.scale(C3) + b2.scale(C7))
    t = ck.check(f"row {r} acc7 first dot", b0.scale(C7) - b1.scale(C5), I32, "i32")
    acc[7] = d(7, t - b3.scale(C1) + b2.scale(C3))
    coef = [ck.check(f"row {r} coefficient {i}", acc[i].floor_shift(16), I16, "i16") for i in range(8)]
    # row_quant: |c| by v_pk_max_i16(c, 0 - c): the negation must stay in the lane
    for i in range(8):
        ck.check(f"row {r} 0 - coefficient {i}", coef[i].scale(-1), I16, "i16")
    return coef


def quant_checks(ck, cmax):
    """row_quant: level = (|c| * iquant + bias * iquant) >> 20 with the WORST table the host can make
    (jpeg_host.cc FinalizeQuantMatrix = src/quantize.cc:123-148): quant 1 .. 255, bias8 <= 255."""
    worst_sum = 0
    for q in range(1, 256):
        iq = ((1 << 16) + q // 2) // q
        if q == 1:
            iq = 0xffff
        for bias8 in (0x80, 0xff):
            bias = (((bias8 * q) << 4) + 128) >> 8
            worst_sum = max(worst_sum, cmax * iq + bias * iq)
            ck.check_const(f"quant q={q} bias*iquant (u32 addend)", bias * iq, U32, "u32") if q in (1, 255) and bias8 == 0xff else None
    ck.check_const("quant |c|*iquant + bias*iquant, worst table (v_mad_u32_u16)", worst_sum, U32, "u32")
    ck.check_const("quant level = sum >> 20, worst table (15-bit magnitude of an entry)", worst_sum >> 20, (0, 0x7fff), "u15")
    ck.check_const("quant |c| as u16 operand", cmax, U16, "u16")


def prove(lo, hi, label):
    ck = Checker(Fraction(lo), Fraction(hi), label)
    x = [[Aff.var(8 * y + c) for c in range(8)] for y in range(8)]
    cols = [col_pass(ck, [x[y][c] for y in range(8)], f"{c}") for c in range(8)]     # cols[c][r]
    cmax = 0
    for r in range(8):
        coef = row_pass(ck, [cols[c][r] for c in range(8)], r)
        for k in coef:
            mn, mx = k.bounds(ck.lo, ck.hi)
            cmax = max(cmax, int(max(-mn, mx)) + 1)
    quant_checks(ck, cmax)
    return ck, cmax


def colour_ranges():
    """luma_pair / cb_sum / cr_sum (scan_device.h): linear in R, G, B -> extremes at the corners of the cube."""
    rows, bad = [], []

    def chk(name, vals, typ, tname):
        mn, mx = min(vals), max(vals)
        rows.append((name, tname, Fraction(mn), Fraction(mx), typ, Fraction(typ[1] - mx if typ[0] == 0 else min(mn - typ[0], typ[1] - mx))))
        if mn < typ[0] or mx > typ[1]:
            bad.append((name, tname, mn, mx))

    corners = [(r, g, b) for r in (0, 255) for g in (0, 255) for b in (0, 255)]
    rnd = 32768 - (128 << 16)
    y32 = [19595 * r + 38469 * g + 7471 * b + rnd for r, g, b in corners]
    chk("luma 32-bit sum (mod 2^32 = two's complement of it)", y32, I32, "i32")
    chk("luma sample = sum >> 16", [v >> 16 for v in y32], (-128, 127), "[-128,127]")
    for n, scale in ((1, 16), (4, 18)):                       # 4:4:4 single pixel; 4:2:0 2x2 sums
        R = [(n * r, n * g, n * b) for r, g, b in corners]
        chk(f"chroma operands R, G sums as i16 (x{n})", [v for t in R for v in t[:2]], I16, "i16")
        chk(f"chroma blue pair halves as u16 (x{n})", [t[2] for t in R], U16, "u16")
        half = 32768 * (n if n == 4 else 1)
        cb = [-11059 * r - 21709 * g + 32768 * b + half for r, g, b in R]
        cr = [32768 * r - 27439 * g - 5329 * b + half for r, g, b in R]
        chk(f"Cb 32-bit sum (x{n})", cb, I32, "i32")
        chk(f"Cr 32-bit sum (x{n})", cr, I32, "i32")
        chk(f"Cb upper half before the packed >> (x{n})", [v >> 16 for v in cb], I16, "i16")
        chk(f"Cr upper half before the packed >> (x{n})", [v >> 16 for v in cr], I16, "i16")
        chk(f"Cb sample (x{n})", [v >> scale for v in cb], (-127, 128), "[-127,128]")
        chk(f"Cr sample (x{n})", [v >> scale for v in cr], (-127, 128), "[-127,128]")
    return rows, bad


# ------------------------------------------------------------------------------------------------
# 2. the device statements with wrap-around semantics (numpy), against the oracle

def _i16(a):
    return np.asarray(a).astype(np.int64).astype(np.uint16).astype(np.int16)     # wrap into the lane


def _sext24(a):
    a = np.asarray(a, np.int64) & 0xffffff
    return np.where(a & 0x800000, a - (1 << 24), a)


def _mul24(a, k):
    """v_mul_i32_i24: low 32 bits of sext24(a) * sext24(k)"""
    p = (_sext24(a) * _sext24(k)) & 0xffffffff
    return np.where(p & 0x80000000, p - (1 << 32), p)


def _pk_mulhi(a, k):
    return _i16((_mul24(np.asarray(a, np.int64), k) >> 16) & 0xffff)


def _wrap32(a):
    a = np.asarray(a, np.int64) & 0xffffffff
    return np.where(a & 0x80000000, a - (1 << 32), a)


def model_fdct(blocks):
    """blocks [n, 8, 8] int16 samples -> [n, 8, 8] coefficients as the device computes them (wrap-around in every
    lane and accumulator exactly where the hardware wraps)."""
    x = [_i16(blocks[:, y, :]) for y in range(8)]                # rows of samples: a lane per column
    A, S = lambda a, b: _i16(a.astype(np.int64) + b), lambda a, b: _i16(a.astype(np.int64) - b)
    SH = lambda a, n: _i16(a.astype(np.int64) << n)
    NOT = lambda a: _i16(~a.astype(np.int64))
    d07, s07 = S(x[0], x[7]), A(x[0], x[7])
    d25, s25 = S(x[2], x[5]), A(x[2], x[5])
    d34, s34 = S(x[3], x[4]), A(x[3], x[4])
    d16, s16 = S(x[1], x[6]), A(x[1], x[6])
    ed, es = S(s07, s34), A(s07, s34)
    fd, fs = S(s16, s25), A(s16, s25)
    a, b = SH(es, 3), SH(fs, 3)
    r = [None] * 8
    r[0], r[4] = A(a, b), S(a, b)
    ed, fd, d34, d07 = SH(ed, 3), SH(fd, 3), SH(d34, 3), SH(d07, 3)
    r[2] = A(_pk_mulhi(fd, 27146), ed)
    r[6] = S(_pk_mulhi(ed, 27146), fd)
    od = _pk_mulhi(S(d16, d25), 23170 << 4)
    os_ = _pk_mulhi(A(d16, d25), 23170 << 4)
    p3, p1 = S(d34, od), A(d34, od)
    p0, p2 = S(d07, os_), A(d07, os_)
    u3 = _pk_mulhi(p3, 65536 - 21746)
    t4 = _pk_mulhi(p0, 65536 - 21746)
    t5 = _pk_mulhi(p2, 13036)
    r[1] = S(_pk_mulhi(p1, 13036), NOT(p2))
    r[3] = A(p0, NOT(u3))
    r[5] = A(p3, t4)
    r[7] = S(t5, p1)
    out = np.zeros(blocks.shape, np.int64)
    for row in range(8):
        C1, C2, C3, C4, C5, C6, C7 = ROW_TABLES[row]
        v = r[row]                                               # [n, 8]: the row's eight column outputs
        a01 = (A(v[:, 0], v[:, 7]), A(v[:, 1], v[:, 6]))
        b01 = (S(v[:, 0], v[:, 7]), S(v[:, 1], v[:, 6]))
        a32 = (A(v[:, 3], v[:, 4]), A(v[:, 2], v[:, 5]))
        b32 = (S(v[:, 3], v[:, 4]), S(v[:, 2], v[:, 5]))
        c13 = (S(a01[0], a32[0]), S(a01[1], a32[1]))
        dot = lambda p, k0, k1, acc=0: _wrap32(p[0].astype(np.int64) * k0 + p[1].astype(np.int64) * k1 + acc)
        acc = [None] * 8
        acc[0] = dot(a32, C4, C4, dot(a01, C4, C4))
        acc[4] = dot(a32, C4, -C4, dot(a01, C4, -C4))
        acc[2] = dot(c13, C2, C6)
        acc[6] = dot(c13, C6, -C2)
        acc[1] = dot(b32, C7, C5, dot(b01, C1, C3))
        acc[3] = dot(b32, -C5, -C1, dot(b01, C3, -C7))
        acc[5] = dot(b32, C3, C7, dot(b01, C5, -C1))
        acc[7] = dot(b32, -C1, C3, dot(b01, C7, -C5))
        for i in range(8):
            out[:, row, i] = _i16(acc[i] >> 16)
    return out.astype(np.int16)


def blocks_from_masks(masks, lo, hi):
    m = np.array([[(mk >> i) & 1 for i in range(64)] for mk in masks], np.int16).reshape(-1, 8, 8)
    return (m * (hi - lo) + lo).astype(np.int16)


def run_model(all_masks):
    from oracle import orc
    o = orc.oracle()
    rng = np.random.default_rng(20260929)
    fails = 0
    n = 0
    for lo, hi in ((-128, 127), (-127, 128)):
        sets = [blocks_from_masks(all_masks, lo, hi)]
        # lattices: per-column / per-row constant blocks and random corners
        col = rng.integers(0, 2, (4000, 1, 8)).repeat(8, 1)
        row = rng.integers(0, 2, (4000, 8, 1)).repeat(8, 2)
        rnd = rng.integers(0, 2, (4000, 8, 8))
        for s in (col, row, rnd, col ^ row):
            sets.append((s * (hi - lo) + lo).astype(np.int16))
        sets.append(rng.integers(lo, hi + 1, (4000, 8, 8)).astype(np.int16))
        for blk in sets:
            got = model_fdct(blk)
            want = o.fdct(blk.reshape(-1, 64).copy()).reshape(-1, 8, 8)
            bad = int((got != want).any(axis=(1, 2)).sum())
            fails += bad
            n += len(blk)
    # the model must also SEE an overflow where there is one: samples -128 .. 128 (the union) wrap c1
    blk = np.full((1, 8, 8), -128, np.int16)
    blk[0, :, 0] = blk[0, :, 7] = 128
    wrapped = (model_fdct(blk) != o.fdct(blk.reshape(-1, 64).copy()).reshape(-1, 8, 8)).any()
    return n, fails, bool(wrapped)


# ------------------------------------------------------------------------------------------------

def fmt_table(ck, top=40):
    rows = sorted(ck.rows, key=lambda r: r[5] / (r[4][1] - r[4][0]))
    out = [f"== samples {ck.lo} .. {ck.hi} ({ck.label}): {len(ck.rows)} checked values, "
           f"{len(ck.bad)} can leave their type; the {top} tightest:"]
    out.append(f"{'value':58s} {'type':>6s} {'min':>14s} {'max':>14s} {'room':>10s}")
    for name, tname, mn, mx, typ, margin in rows[:top]:
        out.append(f"{name:58s} {tname:>6s} {float(mn):14.1f} {float(mx):14.1f} {float(margin):10.1f}")
    return "\n".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write-patterns", action="store_true")
    ap.add_argument("--table", default=None)
    ap.add_argument("--no-model", action="store_true")
    args = ap.parse_args()
    text, ok = [], True
    pattern_masks = {}
    for lo, hi, label in ((-128, 127, "luma; planar and sharp-YUV sources"), (-127, 128, "chroma from RGB")):
        ck, cmax = prove(lo, hi, label)
        text.append(fmt_table(ck))
        text.append(f"largest |coefficient| <= {cmax}")
        ok = ok and not ck.bad
        for b in ck.bad:
            text.append("  CAN WRAP: %s (%s): %.1f .. %.1f" % (b[0], b[1], float(b[2]), float(b[3])))
        # the corners of the values with the least room relative to their type: what the GPU tests paint
        tight = sorted((r for r in ck.rows if r[0] in ck.patterns), key=lambda r: r[5] / (r[4][1] - r[4][0]))
        for name, *_ in tight[:48]:
            pattern_masks.setdefault(name, ck.patterns[name])
    crow, cbad = colour_ranges()
    text.append("== colour conversion, corners of the RGB cube")
    for name, tname, mn, mx, typ, margin in crow:
        text.append(f"{name:58s} {tname:>11s} {float(mn):14.1f} {float(mx):14.1f} {float(margin):10.1f}")
    ok = ok and not cbad
    masks = sorted({m for pair in pattern_masks.values() for m in pair})
    if not args.no_model:
        n, fails, wrapped = run_model(masks)
        text.append(f"== wrap-around model of the device statements against the oracle's fDCT: {n} blocks "
                    f"(extremal corners, column / row / random lattices, random samples), {fails} mismatches; "
                    f"the union range -128 .. 128 is seen to wrap: {wrapped}")
        ok = ok and fails == 0 and wrapped
    text.append("RESULT: " + ("no lane can leave its type" if ok else "FAILED"))
    out = "\n".join(text)
    print(out)
    if args.table:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.table)), exist_ok=True)
        open(os.path.join(ROOT, args.table), "w").write(out + "\n")
    if args.write_patterns:
        js = {"comment": "corner patterns of the tightest lanes of K1's packed fDCT (tools/int16_ranges.py): bit i of a "
                         "mask = sample i (row-major 8x8) at the upper end of its range",
              "patterns": [{"value": k, "max": "%016x" % v[0], "min": "%016x" % v[1]} for k, v in sorted(pattern_masks.items())]}
        json.dump(js, open(os.path.join(ROOT, "tests", "golden", "extremal_patterns.json"), "w"), indent=0)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
