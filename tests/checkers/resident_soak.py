#!/usr/bin/python3
"""Soak of the resident Relax kernels against the oracle (a checker run by hand on the GPU box): random level shapes of 33 ... 129
points per row, sweep counts 1 ... 40, both precisions, plain and from-zero calls, all three forms ("relax3d.resident" 1 / 2,
"relax3d.resident_tile" 0 / 8) interleaved on ONE context, so that exchange-buffer lay-outs, launch epochs and tags of one form
meet those of the others.
    python3 tests/checkers/resident_soak.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O  # noqa: E402
import pde_multigrid_amd as P  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = P.Context(0)
ctx.set_param("relax3d.resident_min", 1)
bad = 0
for c in range(cases):
    n3 = (int(rng.choice([33, 65, 129])), int(rng.choice([9, 17, 33, 65, 129])), int(rng.choice([9, 17, 33, 65, 129])))
    dtype = np.float64 if rng.random() < 0.5 else np.float32
    form, tile = [(1, 0), (1, 8), (2, 0)][int(rng.integers(0, 3))]
    nc = int(rng.choice([1, 2, 3, 4, 5, 7, 12, 40]))
    zero = rng.random() < 0.3
    ctx.set_param("relax3d.resident", form)
    ctx.set_param("relax3d.resident_tile", tile)
    rg = [0, 1, 0, 1, 0, 1] if rng.random() < 0.5 else [-1, 1, 0, 2, 0.5, 3]
    shape = tuple(reversed(n3))
    v, f = rng.uniform(-1, 1, shape).astype(dtype), rng.uniform(-1, 1, shape).astype(dtype)
    if zero:
        v[0] = v[-1] = 0
        v[:, 0] = v[:, -1] = 0
        v[:, :, 0] = v[:, :, -1] = 0
        got = P.ops3dxs.relax_from_zero(ctx, v, f, n3, rg, nc, True)
        want = O.relax3d(n3, rg, np.zeros_like(v), f, nc, dtype=dtype)
    else:
        got = P.ops3dxs.relax(ctx, v, f, n3, rg, nc)
        want = O.relax3d(n3, rg, v, f, nc, dtype=dtype)
    k = ctx.last_relax_kernel()
    ctx.sync()
    U = np.uint64 if dtype == np.float64 else np.uint32
    ok = bool((got.view(U) == want.view(U)).all()) and k.startswith("relax3d_xs_resident")
    bad += not ok
    print("%s %s %s sweeps=%d zero=%d form=%d tile=%d [%s]" % ("ok  " if ok else "FAIL", n3, np.dtype(dtype).name, nc, zero, form, tile, k), flush=True)
print("%d cases, %d failures" % (cases, bad))
ctx.close()
sys.exit(1 if bad else 0)
