"""Known answers of the fp64 BASELINE configurations from the oracle's CPU restatement -- TEST INFRASTRUCTURE.

The reference is fp32 only, so the fp64 oracle is restatement<double> (pinned transitively: restatement<float> is
bit-identical to the compiled reference, tests/test_oracle_vs_ref.py).  A 1025^3 cycle takes minutes and ~47 GB on the
CPU, so its result is hashed once here and the GPU tests / bench.py compare against the committed values
(tests/golden/known_answers_f64.json) instead of re-running the oracle on the GPU box.

    python oracle/gen_known_f64.py [513] [1025]

Per case: `fnv` = oracle.fnv of the finest v after ONE V(2,2) cycle from v = 0 (analytic RHS of Grid3D::InitF,
reference residual semantics), `sum64` / `wsum64` = the position-weighted 64-bit checksum bench.py can evaluate
with numpy alone (checksum64 below), `centre` = the centre value, `block_fnv` = fnv per block of 64 planes
(to localise a mismatch).  InitF depends on libm's sin: valid for the glibc of this image (2.35).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "known_answers_f64.json")
R3 = [0, 1, 0, 1, 0, 1]


def checksum64(a, first_word=0):
    """(sum w_i, sum w_i * (2 i + 1)) mod 2^64 over the words of `a` in memory order (64-bit words for fp64, 32-bit words
    for fp32), i counted from first_word -- bench.py restates this function on its product side"""
    w = np.ascontiguousarray(a).reshape(-1)
    w = w.view(np.uint64) if w.dtype.itemsize == 8 else w.view(np.uint32)
    s1 = np.uint64(0)
    s2 = np.uint64(0)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for i in range(0, w.size, step):
            c = w[i:i + step].astype(np.uint64)
            k = np.arange(first_word + i, first_word + i + c.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            s1 = s1 + c.sum(dtype=np.uint64)
            s2 = s2 + (c * k).sum(dtype=np.uint64)
    return "%016x" % int(s1), "%016x" % int(s2)


def case(n, dtype=np.float64):
    nlev = O.num_grids(n)
    v = O.cycle3d([n] * 3, R3, mode=0, v1=2, v2=2, reps=1, dtype=dtype)
    s1, s2 = checksum64(v)
    return {
        "n": n, "nlevels": nlev, "v1": 2, "v2": 2, "dtype": "f64" if dtype == np.float64 else "f32", "fnv": O.fnv(v),
        "sum64": s1, "wsum64": s2, "centre": float(v[n // 2, n // 2, n // 2]),
        "block_fnv": [O.fnv(v[z:z + 64]) for z in range(0, n, 64)],
    }


def case_levels(n, nlev, dtype):
    """3D, explicit level count (BASELINE configs[2]: 257^3 with 6 levels)"""
    v = O.cycle3d([n] * 3, R3, nlevels=nlev, mode=0, v1=2, v2=2, reps=1, dtype=dtype)
    s1, s2 = checksum64(v)
    return {"n": n, "nlevels": nlev, "v1": 2, "v2": 2, "dtype": "f64" if dtype == np.float64 else "f32", "fnv": O.fnv(v),
            "sum64": s1, "wsum64": s2, "centre": float(v[n // 2, n // 2, n // 2])}


def case2d(n, nlev, dtype):
    """2D Lyapunov (BASELINE configs[1]: 1025^2, 7 levels; the reference driver's A, alfa, N2/LyapunovSolver.cpp:13-28, unit box)"""
    v = O.cycle2d([n] * 2, [0, 1, 0, 1], [-1.0, -2.0, 0.0, -3.0], 2, nlevels=nlev, mode=0, v1=2, v2=2, reps=1, dtype=dtype)
    s1, s2 = checksum64(v)
    return {"n": n, "nlevels": nlev, "v1": 2, "v2": 2, "dtype": "f64" if dtype == np.float64 else "f32", "fnv": O.fnv(v),
            "sum64": s1, "wsum64": s2, "centre": float(v[n // 2, n // 2])}


def secondary(data):
    """the configurations bench.py reports under `secondary` (keys carry the level count and the type)"""
    for dt, t in (("f64", np.float64), ("f32", np.float32)):
        data["3d_n257_vcycle22_6lev_%s" % dt] = case_levels(257, 6, t)
        data["2d_n1025_vcycle22_7lev_%s" % dt] = case2d(1025, 7, t)
        for k in ("3d_n257_vcycle22_6lev_%s" % dt, "2d_n1025_vcycle22_7lev_%s" % dt):
            print(k, data[k]["fnv"], data[k]["centre"], flush=True)


def thesis(data, sizes):
    """the reference's published 3D workload (thesis Fig. 4.4 = BASELINE.md section 1; N3/Poisson3DSolver.cpp:18-20): full multigrid
    with 2 V-cycles per level and 3000 + 3000 sweeps per visit, fp32 = restatement<float> = the compiled reference's bits.
    n = 129 takes about ten minutes on one core: `python oracle/gen_known_f64.py thesis [65] [129]`"""
    for n in sizes:
        v = O.cycle3d([n] * 3, R3, mode=1, v0=2, v1=3000, v2=3000, dtype=np.float32)
        s1, s2 = checksum64(v)
        key = "3d_n%d_fmg_2_3000_3000_f32" % n
        data[key] = {"n": n, "nlevels": O.num_grids(n), "v0": 2, "v1": 3000, "v2": 3000, "dtype": "f32", "fnv": O.fnv(v), "sum64": s1,
                     "wsum64": s2, "centre": float(v[n // 2, n // 2, n // 2])}
        print(key, data[key]["fnv"], data[key]["centre"], flush=True)


def thesis2d(data, sizes):
    """the reference's published 2D workload (thesis Fig. 4.2 = BASELINE.md section 1; N2/LyapunovSolver.cpp:13-31 on [0, 20]^2): full
    multigrid with 2 V-cycles per level and 500 + 500 sweeps per visit, fp32.  `python oracle/gen_known_f64.py thesis2d [1025] [4097]`"""
    for n in sizes:
        v = O.cycle2d([n] * 2, [0, 20, 0, 20], [-1.0, -2.0, 0.0, -3.0], 2, mode=1, v0=2, v1=500, v2=500, dtype=np.float32)
        s1, s2 = checksum64(v)
        key = "2d_n%d_fmg_2_500_500_f32" % n
        data[key] = {"n": n, "nlevels": O.num_grids(n), "v0": 2, "v1": 500, "v2": 500, "dtype": "f32", "fnv": O.fnv(v), "sum64": s1,
                     "wsum64": s2, "centre": float(v[n // 2, n // 2])}
        print(key, data[key]["fnv"], data[key]["centre"], flush=True)


def main():
    """arguments: sizes, each optionally suffixed with the type, e.g. `513 1025 513:f32` (default f64).  fp32 cases are
    restatement<float>, which is bit-identical to the compiled reference (tests/test_oracle_vs_ref.py)"""
    todo = [a.split(":") for a in sys.argv[1:]] or [["513"], ["1025"]]
    data = {}
    if os.path.exists(OUT):
        with open(OUT) as fh:
            data = json.load(fh)
    if todo and todo[0] == ["thesis"]:
        thesis(data, [int(t[0]) for t in todo[1:]] or [65, 129])
        with open(OUT, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
        return
    if todo and todo[0] == ["thesis2d"]:
        thesis2d(data, [int(t[0]) for t in todo[1:]] or [1025, 4097])
        with open(OUT, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
        return
    if todo == [["secondary"]]:
        secondary(data)
        with open(OUT, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)
        return
    for t in todo:
        n, dt = int(t[0]), (t[1] if len(t) > 1 else "f64")
        key = "3d_n%d_vcycle22_%dlev_%s" % (n, O.num_grids(n), dt)
        data[key] = case(n, np.float64 if dt == "f64" else np.float32)
        print(key, data[key]["fnv"], data[key]["centre"], flush=True)
        with open(OUT, "w") as fh:
            json.dump(data, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
