"""ctypes binding of oracle/_ref/libmgref.so -- TEST INFRASTRUCTURE, container-only.

libmgref.so is the UNMODIFIED NOCUDA_TESI reference (compiled by `make -C oracle ref`
from /root/reference) behind the extern "C" driver oracle/ref_shim.cpp.  It exists
only where /root/reference exists.  Used by oracle/gen_golden.py to produce the
fixtures in tests/golden/ and by tests/test_oracle_vs_ref.py to pin the restatement.
Never imported by the product (pde_multigrid_amd/).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libmgref.so")
LIB_PATH_O0 = os.path.join(_HERE, "_ref", "libmgref_O0.so")  # what the reference's own CompileAndLink builds (no -O flag)


def available(opt="O2"):
    return os.path.exists(LIB_PATH if opt == "O2" else LIB_PATH_O0)


def time_vcycle3d(n, nlevels, v1, v2, reps, opt="O2"):
    """seconds of `reps` x MultiGrid3D::VCycle(0, v1, v2) of the compiled reference (fp32, 1 thread); bench.py cpu_baseline"""
    so = C.CDLL(LIB_PATH if opt == "O2" else LIB_PATH_O0)
    so.ref3d_time_vcycle.restype = C.c_double
    return so.ref3d_time_vcycle(C.c_int(n), C.c_int(nlevels), C.c_int(v1), C.c_int(v2), C.c_int(reps))


def time_relax3d(n, sweeps, opt="O2"):
    so = C.CDLL(LIB_PATH if opt == "O2" else LIB_PATH_O0)
    so.ref3d_time_relax.restype = C.c_double
    return so.ref3d_time_relax(C.c_int(n), C.c_int(sweeps))


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _ip(a):
    return (C.c_int * len(a))(*[int(x) for x in a])


def _fp(a):
    return (C.c_float * len(a))(*[float(x) for x in a])


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def csize(n):
    return tuple((int(k) - 1) // 2 + 1 for k in n)


def _shape(n):
    # arrays are indexed [z][y][x] in numpy terms (x fastest), matching idx = x + y*sx + z*sx*sy
    return tuple(int(k) for k in reversed(n))


# ------------------------------------------------------------------ 3D ----
def init3d(n, rng, level=0):
    nl = list(n)
    for _ in range(level):
        nl = list(csize(nl))
    v = np.zeros(_shape(nl), np.float32)
    f = np.zeros(_shape(nl), np.float32)
    lib().ref3d_init(_ip(n), _fp(rng), C.c_int(level), _p(v), _p(f))
    return v, f


def relax3d(n, rng, v, f, ncycles):
    v = _f32(v).copy()
    f = _f32(f)
    lib().ref3d_relax(_ip(n), _fp(rng), _p(v), _p(f), C.c_int(ncycles))
    return v


def residual3d(n, rng, v, f):
    v = _f32(v)
    f = _f32(f)
    r = np.empty_like(v)
    lib().ref3d_residual(_ip(n), _fp(rng), _p(v), _p(f), _p(r))
    return r


def restrict3d(n, fine):
    fine = _f32(fine)
    coarse = np.zeros(_shape(csize(n)), np.float32)
    lib().ref3d_restrict(_ip(n), _p(fine), _p(coarse))
    return coarse


def interpolate3d(n, fine, coarse):
    fine = _f32(fine).copy()
    coarse = _f32(coarse)
    lib().ref3d_interpolate(_ip(n), _p(fine), _p(coarse))
    return fine


def correct3d(n, fine, err):
    fine = _f32(fine).copy()
    err = _f32(err)
    lib().ref3d_apply_correction(_ip(n), _p(fine), _p(err))
    return fine


def set3d(n, grid, value, modify_boundaries):
    grid = _f32(grid).copy()
    lib().ref3d_set(_ip(n), _p(grid), C.c_float(value), C.c_int(int(modify_boundaries)))
    return grid


def cycle3d(n, rng, nlevels=0, mode=0, v0=1, v1=2, v2=2, reps=1, v=None, f=None):
    v = _f32(v) if v is not None else None
    f = _f32(f) if f is not None else None
    out = np.empty(_shape(n), np.float32)
    lib().ref3d_cycle(_ip(n), _fp(rng), C.c_int(nlevels), C.c_int(mode), C.c_int(v0),
                      C.c_int(v1), C.c_int(v2), C.c_int(reps), _p(v), _p(f), _p(out))
    return out


# ------------------------------------------------------------------ 2D ----
def init2d(n, rng, A, alfa, level=0):
    nl = list(n)
    for _ in range(level):
        nl = list(csize(nl))
    v = np.zeros(_shape(nl), np.float32)
    f = np.zeros(_shape(nl), np.float32)
    lib().ref2d_init(_ip(n), _fp(rng), _fp(A), C.c_int(alfa), C.c_int(level), _p(v), _p(f))
    return v, f


def relax2d(n, rng, A, alfa, v, f, ncycles):
    v = _f32(v).copy()
    f = _f32(f)
    lib().ref2d_relax(_ip(n), _fp(rng), _fp(A), C.c_int(alfa), _p(v), _p(f), C.c_int(ncycles))
    return v


def residual2d(n, rng, A, alfa, v, f):
    v = _f32(v)
    f = _f32(f)
    r = np.empty_like(v)
    lib().ref2d_residual(_ip(n), _fp(rng), _fp(A), C.c_int(alfa), _p(v), _p(f), _p(r))
    return r


def restrict2d(n, fine):
    fine = _f32(fine)
    coarse = np.zeros(_shape(csize(n)), np.float32)
    lib().ref2d_restrict(_ip(n), _p(fine), _p(coarse))
    return coarse


def interpolate2d(n, fine, coarse):
    fine = _f32(fine).copy()
    coarse = _f32(coarse)
    lib().ref2d_interpolate(_ip(n), _p(fine), _p(coarse))
    return fine


def correct2d(n, fine, err):
    fine = _f32(fine).copy()
    err = _f32(err)
    lib().ref2d_apply_correction(_ip(n), _p(fine), _p(err))
    return fine


def set2d(n, grid, value, modify_boundaries):
    grid = _f32(grid).copy()
    lib().ref2d_set(_ip(n), _p(grid), C.c_float(value), C.c_int(int(modify_boundaries)))
    return grid


def cycle2d(n, rng, A, alfa, nlevels=0, mode=0, v0=1, v1=2, v2=2, reps=1, v=None, f=None):
    v = _f32(v) if v is not None else None
    f = _f32(f) if f is not None else None
    out = np.empty(_shape(n), np.float32)
    lib().ref2d_cycle(_ip(n), _fp(rng), _fp(A), C.c_int(alfa), C.c_int(nlevels), C.c_int(mode),
                      C.c_int(v0), C.c_int(v1), C.c_int(v2), C.c_int(reps), _p(v), _p(f), _p(out))
    return out


# ------------------------------------------------------------------ 1D ----
def init1d(n, rng, level=0):
    nl = int(n)
    for _ in range(level):
        nl = (nl - 1) // 2 + 1
    v = np.zeros(nl, np.float32)
    f = np.zeros(nl, np.float32)
    lib().ref1d_init(C.c_int(n), _fp(rng), C.c_int(level), _p(v), _p(f))
    return v, f


def relax1d(n, rng, v, f, ncycles):
    v = _f32(v).copy()
    f = _f32(f)
    lib().ref1d_relax(C.c_int(n), _fp(rng), _p(v), _p(f), C.c_int(ncycles))
    return v


def residual1d(n, rng, v, f):
    v = _f32(v)
    f = _f32(f)
    r = np.empty_like(v)
    lib().ref1d_residual(C.c_int(n), _fp(rng), _p(v), _p(f), _p(r))
    return r


def restrict1d(n, fine):
    fine = _f32(fine)
    coarse = np.zeros((int(n) - 1) // 2 + 1, np.float32)
    lib().ref1d_restrict(C.c_int(n), _p(fine), _p(coarse))
    return coarse


def interpolate1d(n, fine, coarse):
    fine = _f32(fine).copy()
    coarse = _f32(coarse)
    lib().ref1d_interpolate(C.c_int(n), _p(fine), _p(coarse))
    return fine


def correct1d(n, fine, err):
    fine = _f32(fine).copy()
    err = _f32(err)
    lib().ref1d_apply_correction(C.c_int(n), _p(fine), _p(err))
    return fine


def set1d(n, grid, value, modify_boundaries):
    grid = _f32(grid).copy()
    lib().ref1d_set(C.c_int(n), _p(grid), C.c_float(value), C.c_int(int(modify_boundaries)))
    return grid


def cycle1d(n, rng, nlevels=0, mode=0, v0=1, v1=2, v2=2, reps=1, v=None, f=None):
    v = _f32(v) if v is not None else None
    f = _f32(f) if f is not None else None
    out = np.empty(int(n), np.float32)
    lib().ref1d_cycle(C.c_int(n), _fp(rng), C.c_int(nlevels), C.c_int(mode), C.c_int(v0),
                      C.c_int(v1), C.c_int(v2), C.c_int(reps), _p(v), _p(f), _p(out))
    return out


# ---------------------------------------------------------------- hash ----
def fnv_bits(a):
    """64-bit FNV-1a-style hash over the 32-bit (or 64-bit, folded) patterns of `a` in
    memory order: h = 0xcbf29ce484222325; per word h = (h ^ bits) * 0x100000001b3
    (SURVEY.md section 8c)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        w = a.view(np.uint32).ravel().astype(np.uint64)
    elif a.dtype == np.float64:
        w = a.view(np.uint32).ravel().astype(np.uint64)  # low word, high word, in memory order
    else:
        raise TypeError(a.dtype)
    h = 0xCBF29CE484222325
    P = 0x100000001B3
    M = (1 << 64) - 1
    for x in w.tolist():
        h = ((h ^ x) * P) & M
    return "%016x" % h
