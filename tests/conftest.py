import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the hot path (oracle/sjpeg_oracle.c) -- the checker, never the product."""
    from oracle import orc
    return orc.oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference build (oracle/_ref); only in environments where it was built."""
    from oracle import refso
    if not refso.available():
        pytest.skip("oracle/_ref/libsjpeg_ref.so not built (needs /root/reference)")
    return refso.ref()


@pytest.fixture(scope="session")
def golden_small():
    z = np.load(os.path.join(GOLDEN, "small.npz"))
    return {k: z[k].tobytes() for k in z.files}


@pytest.fixture(scope="session")
def digests():
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def img128():
    return np.fromfile(os.path.join(GOLDEN, "test128.rgb"), np.uint8).reshape(128, 128, 3)


MODES = {"420": 1, "444": 3, "400": 4}


def golden_input(key):
    """Regenerates the input of a small.npz key; returns (img, mode, quality, method)."""
    from oracle import synth
    name, dims, mname, q, m = key.split("|")
    w, h = (int(v) for v in dims.split("x"))
    if name == "test128":
        img = np.fromfile(os.path.join(GOLDEN, "test128.rgb"), np.uint8).reshape(128, 128, 3)
    else:
        gen = synth.g_struct if name == "struct" else synth.g_noise
        img = gen(w, h, 7654321 + w)
    return img, MODES[mname], float(q[1:]), int(m[1:])
