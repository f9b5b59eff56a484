#!/usr/bin/python3
"""One-off fuzz of the 3D / 2D cycle drivers against the oracle (test infrastructure use of oracle/: this is a checker, run
by hand on the GPU box, not part of the suites): random anisotropic 2^k+1 shapes, sweep counts, level counts, residual
modes, both precisions, V-cycles and FMG, boxes with and without power-of-two spacings.

    python3 tests/checkers/fuzz_cycles.py [cases] [seed]
MGX_PARAMS=name=value,... sets library parameters first (e.g. rr3d.black=2,relax3d.resident_min=1: the fused way down and
the resident Relax kernel on every level that has the geometry); MGX_FUZZ_SWEEPS=n: sweep counts up to n - 1 (default 4); MGX_FUZZ_WIDE=1: extents of 513 and 1025 points too.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
import pde_multigrid_amd as P  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = P.Context(0)
for kv in filter(None, os.environ.get("MGX_PARAMS", "").split(",")):
    ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))
SWEEPS = int(os.environ.get("MGX_FUZZ_SWEEPS", "4"))
SZ = [3, 5, 9, 17, 33, 65, 129, 257]
if os.environ.get("MGX_FUZZ_WIDE"):  # rows of 513 / 1025 points too (the fp32 two-pair kernels take rows of >= 513 points)
    SZ = SZ + [513, 513, 1025]
bad = 0
for c in range(cases):
    dim = 3 if rng.random() < 0.7 else 2
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    while True:
        n = [int(rng.choice(SZ)) for _ in range(dim)]
        if np.prod(n) <= (9e6 if dim == 3 else 1.1e6) and max(n) >= 9:
            break
    box = []
    for d in range(dim):
        a = float(rng.choice([0.0, -1.0, 0.5]))
        box += [a, a + float(rng.choice([1.0, 2.0, 1.5, 20.0]))]
    v1, v2 = int(rng.integers(0, SWEEPS)), int(rng.integers(0, SWEEPS))
    maxlev = O.num_grids(min(n))
    nlev = int(rng.integers(1, maxlev + 1)) if rng.random() < 0.5 else 0
    fmg = rng.random() < 0.3
    reps = 1 if fmg else int(rng.integers(1, 3))
    v = rng.uniform(-1, 1, O.shape(n)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n)).astype(dtype)
    if dim == 3:
        mode = int(rng.integers(0, 2))
        mg = P.MultiGrid3D(ctx, n, box, dtype, nlevels=nlev, residual_mode=mode)
        want = O.cycle3d(n, box, nlevels=nlev, mode=1 if fmg else 0, v0=2, v1=v1, v2=v2, reps=reps, v=v, f=f, residual_mode=mode, dtype=dtype)
    else:
        mode = 0
        A = [-1.0, -2.0, 0.0, -3.0]
        mg = P.MultiGrid2D(ctx, n, box, A, 2, dtype, nlevels=nlev)
        want = O.cycle2d(n, box, A, 2, nlevels=nlev, mode=1 if fmg else 0, v0=2, v1=v1, v2=v2, reps=reps, v=v, f=f, dtype=dtype)
    mg.upload_v(0, v)
    mg.upload_f(0, f)
    if fmg:
        mg.FullMultiGridVCycle(0, 2, v1, v2)
    else:
        for _ in range(reps):
            mg.VCycle(0, v1, v2)
    got = mg.download_v(0)
    mg.close()
    # a NaN BORN on the device (0 / 0 where the 2D operator's denominator vanishes on boxes with negative coordinates) is
    # +NaN on gfx950 and -NaN on x86: the same positions must be NaN, every other word must have the same bits
    nan = np.isnan(want)
    ok = np.array_equal(np.isnan(got), nan) and got[~nan].tobytes() == want[~nan].tobytes()
    bad += not ok
    print("%s %dD n=%s box=%s %s nlev=%d V(%d,%d)x%d %s mode=%d" % ("ok  " if ok else "FAIL", dim, n, box, np.dtype(dtype).name, nlev, v1, v2, reps,
                                                                 "fmg" if fmg else "v", mode), flush=True)
print("%d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
