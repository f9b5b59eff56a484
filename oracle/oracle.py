"""ctypes binding of oracle/libmgoracle.so -- TEST INFRASTRUCTURE.

The CPU restatement (oracle/mg_oracle.hpp) of the reference's multigrid operators, in
float (pinned bit-exact to the compiled reference and to tests/golden/) and double
(the fp64 oracle).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module; the product package never does.

numpy arrays are indexed [z, y, x] (x fastest in memory), i.e. the reference's
idx = x + y*sx + z*sx*sy; sizes `n` are given as (sx, sy, sz) like the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmgoracle.so")

REF_COMPAT, CORRECT = 0, 1
_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, s))
                                             for s in ("mg_oracle.cpp", "mg_oracle.hpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"] + (["-B"] if force else []))


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.mgo_fnv_words32.restype = C.c_uint64
        _lib.mgo_fnv_words32.argtypes = [C.c_void_p, C.c_size_t]
        for nm in ("mgo3d_time_relax_f32", "mgo3d_time_relax_f64"):
            getattr(_lib, nm).restype = C.c_double
            getattr(_lib, nm).argtypes = [C.c_int, C.c_int]
        for nm in ("mgo3d_time_vcycle_f32", "mgo3d_time_vcycle_f64"):
            getattr(_lib, nm).restype = C.c_double
            getattr(_lib, nm).argtypes = [C.c_int] * 5
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(dtype)


def _ip(a):
    return (C.c_int * len(a))(*[int(x) for x in a])


def _rp(a, ct):
    return (ct * len(a))(*[float(x) for x in a])


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p) if arr is not None else None


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def csize(n):
    return tuple((int(k) - 1) // 2 + 1 for k in n)


def shape(n):
    return tuple(int(k) for k in reversed(n))


def num_grids(min_size):
    return lib().mgo_num_grids(int(min_size))


def fnv(a):
    a = np.ascontiguousarray(a)
    return "%016x" % lib().mgo_fnv_words32(_p(a), a.nbytes // 4)


def _fn(name, dtype):
    s, ct = _sfx(dtype)
    return getattr(lib(), "%s_%s" % (name, s)), ct


# ------------------------------------------------------------------ 3D ----
def init3d(n, rng, level=0, dtype=np.float32):
    fn, ct = _fn("mgo3d_init", dtype)
    nl = list(n)
    for _ in range(level):
        nl = list(csize(nl))
    v = np.zeros(shape(nl), dtype)
    f = np.zeros(shape(nl), dtype)
    fn(_ip(n), _rp(rng, ct), C.c_int(level), _p(v), _p(f))
    return v, f


def relax3d(n, rng, v, f, ncycles, dtype=np.float32):
    fn, ct = _fn("mgo3d_relax", dtype)
    v = _arr(v, dtype).copy()
    f = _arr(f, dtype)
    fn(_ip(n), _rp(rng, ct), _p(v), _p(f), C.c_int(ncycles))
    return v


def relax_colour3d(n, rng, v, f, colour, dtype=np.float32):
    """one colour pass (0 = red, 1 = black) over all interior points"""
    fn, ct = _fn("mgo3d_relax_colour", dtype)
    v = _arr(v, dtype).copy()
    f = _arr(f, dtype)
    fn(_ip(n), _rp(rng, ct), _p(v), _p(f), C.c_int(colour))
    return v


def jacobi3d(n, rng, v, f, omega, ncycles, dtype=np.float32):
    fn, ct = _fn("mgo3d_jacobi", dtype)
    v = _arr(v, dtype).copy()
    f = _arr(f, dtype)
    fn(_ip(n), _rp(rng, ct), _p(v), _p(f), ct(omega), C.c_int(ncycles))
    return v


def jacobi2d(n, rng, A, alfa, v, f, omega, ncycles, dtype=np.float32):
    fn, ct = _fn("mgo2d_jacobi", dtype)
    v = _arr(v, dtype).copy()
    f = _arr(f, dtype)
    fn(_ip(n), _rp(rng, ct), _rp(A, ct), C.c_int(alfa), _p(v), _p(f), ct(omega), C.c_int(ncycles))
    return v


def residual3d(n, rng, v, f, mode=REF_COMPAT, dtype=np.float32):
    fn, ct = _fn("mgo3d_residual", dtype)
    v = _arr(v, dtype)
    f = _arr(f, dtype)
    r = np.empty_like(v)
    fn(_ip(n), _rp(rng, ct), _p(v), _p(f), _p(r), C.c_int(mode))
    return r


def restrict3d(n, fine, dtype=np.float32):
    fn, _ = _fn("mgo3d_restrict", dtype)
    fine = _arr(fine, dtype)
    coarse = np.zeros(shape(csize(n)), dtype)
    fn(_ip(n), _p(fine), _p(coarse))
    return coarse


def interpolate3d(n, fine, coarse, dtype=np.float32):
    fn, _ = _fn("mgo3d_interpolate", dtype)
    fine = _arr(fine, dtype).copy()
    coarse = _arr(coarse, dtype)
    fn(_ip(n), _p(fine), _p(coarse))
    return fine


def correct3d(n, fine, err, dtype=np.float32):
    fn, _ = _fn("mgo3d_apply_correction", dtype)
    fine = _arr(fine, dtype).copy()
    err = _arr(err, dtype)
    fn(_ip(n), _p(fine), _p(err))
    return fine


def set3d(n, grid, value, modify_boundaries, dtype=np.float32):
    fn, ct = _fn("mgo3d_set", dtype)
    grid = _arr(grid, dtype).copy()
    fn(_ip(n), _p(grid), ct(value), C.c_int(int(modify_boundaries)))
    return grid


def cycle3d(n, rng, nlevels=0, mode=0, v0=1, v1=2, v2=2, reps=1, v=None, f=None,
            residual_mode=REF_COMPAT, dtype=np.float32):
    fn, ct = _fn("mgo3d_cycle", dtype)
    v = _arr(v, dtype) if v is not None else None
    f = _arr(f, dtype) if f is not None else None
    out = np.empty(shape(n), dtype)
    fn(_ip(n), _rp(rng, ct), C.c_int(nlevels), C.c_int(mode), C.c_int(v0), C.c_int(v1), C.c_int(v2),
       C.c_int(reps), _p(v), _p(f), _p(out), C.c_int(residual_mode))
    return out


# ------------------------------------------------------------------ 2D ----
def init2d(n, rng, level=0, dtype=np.float32):
    fn, ct = _fn("mgo2d_init", dtype)
    nl = list(n)
    for _ in range(level):
        nl = list(csize(nl))
    v = np.zeros(shape(nl), dtype)
    f = np.zeros(shape(nl), dtype)
    fn(_ip(n), _rp(rng, ct), C.c_int(level), _p(v), _p(f))
    return v, f


def relax2d(n, rng, A, alfa, v, f, ncycles, dtype=np.float32):
    fn, ct = _fn("mgo2d_relax", dtype)
    v = _arr(v, dtype).copy()
    f = _arr(f, dtype)
    fn(_ip(n), _rp(rng, ct), _rp(A, ct), C.c_int(alfa), _p(v), _p(f), C.c_int(ncycles))
    return v


def residual2d(n, rng, A, alfa, v, f, dtype=np.float32):
    fn, ct = _fn("mgo2d_residual", dtype)
    v = _arr(v, dtype)
    f = _arr(f, dtype)
    r = np.empty_like(v)
    fn(_ip(n), _rp(rng, ct), _rp(A, ct), C.c_int(alfa), _p(v), _p(f), _p(r))
    return r


def restrict2d(n, fine, dtype=np.float32):
    fn, _ = _fn("mgo2d_restrict", dtype)
    fine = _arr(fine, dtype)
    coarse = np.zeros(shape(csize(n)), dtype)
    fn(_ip(n), _p(fine), _p(coarse))
    return coarse


def interpolate2d(n, fine, coarse, dtype=np.float32):
    fn, _ = _fn("mgo2d_interpolate", dtype)
    fine = _arr(fine, dtype).copy()
    coarse = _arr(coarse, dtype)
    fn(_ip(n), _p(fine), _p(coarse))
    return fine


def correct2d(n, fine, err, dtype=np.float32):
    fn, _ = _fn("mgo2d_apply_correction", dtype)
    fine = _arr(fine, dtype).copy()
    err = _arr(err, dtype)
    fn(_ip(n), _p(fine), _p(err))
    return fine


def set2d(n, grid, value, modify_boundaries, dtype=np.float32):
    fn, ct = _fn("mgo2d_set", dtype)
    grid = _arr(grid, dtype).copy()
    fn(_ip(n), _p(grid), ct(value), C.c_int(int(modify_boundaries)))
    return grid


def cycle2d(n, rng, A, alfa, nlevels=0, mode=0, v0=1, v1=2, v2=2, reps=1, v=None, f=None,
            dtype=np.float32):
    fn, ct = _fn("mgo2d_cycle", dtype)
    v = _arr(v, dtype) if v is not None else None
    f = _arr(f, dtype) if f is not None else None
    out = np.empty(shape(n), dtype)
    fn(_ip(n), _rp(rng, ct), _rp(A, ct), C.c_int(alfa), C.c_int(nlevels), C.c_int(mode), C.c_int(v0),
       C.c_int(v1), C.c_int(v2), C.c_int(reps), _p(v), _p(f), _p(out))
    return out


# ------------------------------------------------------------------ 1D ----
def init1d(n, rng, level=0, dtype=np.float32):
    fn, ct = _fn("mgo1d_init", dtype)
    nl = int(n)
    for _ in range(level):
        nl = (nl - 1) // 2 + 1
    v = np.zeros(nl, dtype)
    f = np.zeros(nl, dtype)
    fn(C.c_int(n), _rp(rng, ct), C.c_int(level), _p(v), _p(f))
    return v, f


def relax1d(n, rng, v, f, ncycles, dtype=np.float32):
    fn, ct = _fn("mgo1d_relax", dtype)
    v = _arr(v, dtype).copy()
    f = _arr(f, dtype)
    fn(C.c_int(n), _rp(rng, ct), _p(v), _p(f), C.c_int(ncycles))
    return v


def residual1d(n, rng, v, f, dtype=np.float32):
    fn, ct = _fn("mgo1d_residual", dtype)
    v = _arr(v, dtype)
    f = _arr(f, dtype)
    r = np.empty_like(v)
    fn(C.c_int(n), _rp(rng, ct), _p(v), _p(f), _p(r))
    return r


def restrict1d(n, fine, dtype=np.float32):
    fn, _ = _fn("mgo1d_restrict", dtype)
    fine = _arr(fine, dtype)
    coarse = np.zeros((int(n) - 1) // 2 + 1, dtype)
    fn(C.c_int(n), _p(fine), _p(coarse))
    return coarse


def interpolate1d(n, fine, coarse, dtype=np.float32):
    fn, _ = _fn("mgo1d_interpolate", dtype)
    fine = _arr(fine, dtype).copy()
    coarse = _arr(coarse, dtype)
    fn(C.c_int(n), _p(fine), _p(coarse))
    return fine


def correct1d(n, fine, err, dtype=np.float32):
    fn, _ = _fn("mgo1d_apply_correction", dtype)
    fine = _arr(fine, dtype).copy()
    err = _arr(err, dtype)
    fn(C.c_int(n), _p(fine), _p(err))
    return fine


def set1d(n, grid, value, modify_boundaries, dtype=np.float32):
    fn, ct = _fn("mgo1d_set", dtype)
    grid = _arr(grid, dtype).copy()
    fn(C.c_int(n), _p(grid), ct(value), C.c_int(int(modify_boundaries)))
    return grid


def cycle1d(n, rng, nlevels=0, mode=0, v0=1, v1=2, v2=2, reps=1, v=None, f=None, dtype=np.float32):
    fn, ct = _fn("mgo1d_cycle", dtype)
    v = _arr(v, dtype) if v is not None else None
    f = _arr(f, dtype) if f is not None else None
    out = np.empty(int(n), dtype)
    fn(C.c_int(n), _rp(rng, ct), C.c_int(nlevels), C.c_int(mode), C.c_int(v0), C.c_int(v1), C.c_int(v2),
       C.c_int(reps), _p(v), _p(f), _p(out))
    return out


# ------------------------------------------------------------ cpu timers ---
LIB_PATH_O0 = os.path.join(_HERE, "libmgoracle_O0.so")  # the restatement without optimisation (the reference's own build uses no -O flag)


def _timer_lib(opt):
    if opt == "O2":
        return lib()
    build()
    so = C.CDLL(LIB_PATH_O0)
    for nm in ("mgo3d_time_relax_f32", "mgo3d_time_relax_f64", "mgo3d_time_vcycle_f32", "mgo3d_time_vcycle_f64"):
        getattr(so, nm).restype = C.c_double
    return so


def time_relax3d(n, sweeps, dtype=np.float64, opt="O2"):
    s, _ = _sfx(dtype)
    return getattr(_timer_lib(opt), "mgo3d_time_relax_" + s)(C.c_int(int(n)), C.c_int(int(sweeps)))


def time_vcycle3d(n, nlevels, v1, v2, reps, dtype=np.float64, opt="O2"):
    s, _ = _sfx(dtype)
    return getattr(_timer_lib(opt), "mgo3d_time_vcycle_" + s)(C.c_int(int(n)), C.c_int(int(nlevels)), C.c_int(int(v1)),
                                                               C.c_int(int(v2)), C.c_int(int(reps)))
