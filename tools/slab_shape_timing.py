#!/usr/bin/python3
"""One GPU's share of the 1025^3 problem as a stand-alone grid: 1025 x 1025 x (1024/N + 1) points, V(2,2) cycle time
(no exchanges) against 1/N of the full 1025^3 cycle.  A sanity check of the kernels on thin slabs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

ctx = P.Context(0)
for nz in (129, 257, 513):
    mg = P.MultiGrid3D(ctx, [1025, 1025, nz], [0, 1, 0, 1, 0, 1], np.float64)
    mg.VCycle(0, 2, 2)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        mg.VCycle(0, 2, 2)
    ctx.sync()
    t = (time.perf_counter() - t0) / 5
    lups, n = 0, [1025, 1025, nz]
    for _ in range(mg.numGrids):
        lups += 4 * (n[0] - 2) * (n[1] - 2) * (n[2] - 2)
        n = [(k - 1) // 2 + 1 for k in n]
    print("1025 x 1025 x %d (share of N = %d): %.3f ms per V(2,2) cycle, %d levels, %.1f GLUPS" % (
        nz, 1024 // (nz - 1), t * 1e3, mg.numGrids, lups / t / 1e9))
    mg.close()
