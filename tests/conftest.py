"""pytest configuration: `gpu` marker, import paths, shared helpers."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits_equal(a, b):
    """Bit-pattern equality (NaN-safe): the parity bar for fp32 and the expected outcome for fp64."""
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    w = np.uint32 if a.dtype == np.float32 else np.uint64
    return bool(np.array_equal(a.view(w), b.view(w)))


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm((a - b).ravel())
    n = np.linalg.norm(b.ravel())
    return d / n if n > 0 else d


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as fh:
        return json.load(fh)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


@pytest.fixture(scope="session", autouse=True)
def _lds_poison(request):
    """GPU runs: every kernel launch of libmgx is preceded by a launch that fills every CU's LDS with NaN patterns
    (mgx_test_set_lds_poison), so a kernel that reads an LDS word before writing it fails its parity test for certain
    instead of depending on the previous launch's leftovers.  MGX_POISON_LDS=0 turns it off."""
    if os.environ.get("MGX_POISON_LDS", "1") == "0" or "not gpu" in (request.config.getoption("-m") or ""):
        yield
        return
    try:
        import pde_multigrid_amd as P
        import ctypes
        cnt = ctypes.c_int(0)
        if P.lib.mgx_device_count(ctypes.byref(cnt)) != 0 or cnt.value <= 0:
            yield
            return
    except Exception:
        yield
        return
    P.lib.mgx_test_set_lds_poison(1)
    yield
    P.lib.mgx_test_set_lds_poison(0)
