"""The range proof of K1's packed int16 / 24-bit arithmetic (tools/int16_ranges.py) as a test: no lane of
fdct_col8_pk / fdct_row8_pk / row_quant / the colour conversion can leave its type for any picture, the wrap-around
model of the device statements equals the oracle's fDCT on the extremal corners, and the committed pattern fixture
is what the proof produces.  CPU only (the GPU side: tests/test_gpu_parity.py::test_extremal_patterns_*)."""
import importlib.util
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("int16_ranges", os.path.join(ROOT, "tools", "int16_ranges.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_lane_can_leave_its_type():
    t = _tool()
    tight = {}
    for lo, hi in ((-128, 127), (-127, 128)):
        ck, cmax = t.prove(lo, hi, "")
        assert ck.bad == [], ck.bad[:3]
        assert len(ck.rows) > 800 and cmax <= 16386
        tight[(lo, hi)] = max(float(r[3]) for r in ck.rows if r[1] == "i16")
    # the knife edge is where the proof says it is: c1 = a0 - a3 of row 0, 127 below the lane's limit
    assert tight == {(-128, 127): 32640.0, (-127, 128): 32640.0}
    rows, bad = t.colour_ranges()
    assert bad == []
    # the union of the two ranges does NOT fit (why the proof is run per range): 8 * 16 * 256 = 32768
    ck, _ = t.prove(-128, 128, "")
    assert any("c1=a0-a3" in b[0] for b in ck.bad)


def test_wraparound_model_equals_the_oracle_on_the_extremal_corners():
    t = _tool()
    pats = json.load(open(os.path.join(ROOT, "tests", "golden", "extremal_patterns.json")))["patterns"]
    masks = sorted({int(p[k], 16) for p in pats for k in ("max", "min")})
    n, fails, wrapped = t.run_model(masks)
    assert n > 40000 and fails == 0
    assert wrapped          # the model is able to see a wrap: it does for samples -128 .. 128


def test_committed_patterns_are_what_the_proof_produces():
    t = _tool()
    want = {}
    for lo, hi in ((-128, 127), (-127, 128)):
        ck, _ = t.prove(lo, hi, "")
        tight = sorted((r for r in ck.rows if r[0] in ck.patterns), key=lambda r: r[5] / (r[4][1] - r[4][0]))
        for name, *_ in tight[:48]:
            want.setdefault(name, ck.patterns[name])
    pats = json.load(open(os.path.join(ROOT, "tests", "golden", "extremal_patterns.json")))["patterns"]
    got = {p["value"]: (int(p["max"], 16), int(p["min"], 16)) for p in pats}
    assert got == want
