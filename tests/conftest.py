import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library honours its XL_EXP_* tuning variables (and XL_LIBRARY_PATH) only next to XL_TESTING=1: the tests force plans through them
# (monkeypatch.setenv("XL_EXP_POLY", ...)); a plain process ignores them (tests/test_option_docs.py checks that side).
os.environ.setdefault("XL_TESTING", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def trunc1e4(a):
    """The reference's float comparison (test/utils.c:176-196): (int32_t)(x * 10000) with x float32."""
    a = np.asarray(a, dtype=np.float32)
    return (a * np.float32(10000)).astype(np.int32)


def assert_ref_cf32(expected_flat, actual_c64):
    """assert_cf32 of test/utils.c:176-182: sizes equal, truncated x10000 equality on re and im."""
    exp = np.asarray(expected_flat, dtype=np.float32)
    act = np.asarray(actual_c64, dtype=np.complex64).view(np.float32)
    assert exp.size == act.size, (exp.size, act.size)
    te, ta = trunc1e4(exp), trunc1e4(act)
    bad = np.nonzero(te != ta)[0]
    assert bad.size == 0, f"truncated mismatch at flat idx {bad[:8]}: exp {exp[bad[:8]]} act {act[bad[:8]]}"


@pytest.fixture(scope="session")
def ref_vectors():
    with open(os.path.join(GOLDEN, "ref_test_vectors.json")) as f:
        return json.load(f)


def load_live(name):
    return np.load(os.path.join(GOLDEN, f"live_{name}.npz"))


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))
