#!/usr/bin/python3
"""2D Lyapunov V(2,2) timing (BASELINE.json configs[1] and larger): cache-resident kernels (fuse = 2) against the
launch-per-colour-pass path (fuse = 1), each eager and as a HIP graph.
    python tools/time_2d.py [sizes ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

ctx = P.Context(0)
sizes = [int(a) for a in sys.argv[1:]] or [1025, 2049, 4097]  # MGX_PARAMS=name=value,... sets context parameters


def timed(fn, reps):
    for _ in range(3):
        fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def lups(n, nlev):
    tot, s = 0, n
    for _ in range(nlev):
        tot += (s - 2) ** 2
        s = (s - 1) // 2 + 1
    return 4 * tot


for k, v in [a.split("=") for a in os.environ.get("MGX_PARAMS", "").split(",") if a]:
    ctx.set_param(k, int(v))
for n in sizes:
    nlev = 7 if n == 1025 else 0
    for dtype, name in ((np.float64, "f64"), (np.float32, "f32")):
        for fuse in (2, 1):
            mg = P.MultiGrid2D(ctx, [n] * 2, [0, 1, 0, 1], [-1, -2, 0, -3], 2, dtype, nlevels=nlev, fuse=fuse)
            t = timed(lambda: mg.VCycle(0, 2, 2), 200)
            mg.use_graph = True
            tg = timed(lambda: mg.VCycle(0, 2, 2), 200)
            print("2D %d^2 %d-level V(2,2) %s fuse=%d: %.4f ms eager, %.4f ms HIP graph  (%.1f GLUPS)"
                  % (n, mg.numGrids, name, fuse, t * 1e3, tg * 1e3, lups(n, mg.numGrids) / min(t, tg) / 1e9))
            mg.close()
