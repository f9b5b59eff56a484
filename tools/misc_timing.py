#!/usr/bin/python3
"""Timing of the secondary configurations (BASELINE.json configs[1], [2]; fp32; FMG) on one MI355X."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

ctx = P.Context(0)
R3 = [0, 1, 0, 1, 0, 1]


def timed(fn, reps):
    fn()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.sync()
    return (time.perf_counter() - t0) / reps


def lups(n, nlev, dim):
    tot, s = 0, n
    for _ in range(nlev):
        tot += (s - 2) ** dim
        s = (s - 1) // 2 + 1
    return 4 * tot


for dtype, name in ((np.float64, "f64"), (np.float32, "f32")):
    mg = P.MultiGrid3D(ctx, [257] * 3, R3, dtype, nlevels=6)
    t = timed(lambda: mg.VCycle(0, 2, 2), 20)
    print("3D 257^3 6-level V(2,2) %s: %.3f ms  %.1f GLUPS" % (name, t * 1e3, lups(257, 6, 3) / t / 1e9))
    mg.close()
    mg = P.MultiGrid3D(ctx, [513] * 3, R3, dtype)
    t = timed(lambda: mg.VCycle(0, 2, 2), 10)
    print("3D 513^3 9-level V(2,2) %s: %.3f ms  %.1f GLUPS" % (name, t * 1e3, lups(513, 9, 3) / t / 1e9))
    mg.use_graph = True
    t = timed(lambda: mg.VCycle(0, 2, 2), 10)
    print("3D 513^3 9-level V(2,2) %s, HIP graph replay: %.3f ms  %.1f GLUPS" % (name, t * 1e3, lups(513, 9, 3) / t / 1e9))
    mg.use_graph = False
    t = timed(lambda: mg.Relax(0, 2), 10)
    print("3D 513^3 smoother %s: %.4f ms/sweep  %.1f GLUPS  %.0f GB/s algorithmic" % (
        name, t / 2 * 1e3, 511 ** 3 / (t / 2) / 1e9, 3 * np.dtype(dtype).itemsize * 511 ** 3 / (t / 2) / 1e9))
    mg.close()
    mg = P.MultiGrid2D(ctx, [1025] * 2, [0, 1, 0, 1], [-1, -2, 0, -3], 2, dtype, nlevels=7)
    t = timed(lambda: mg.VCycle(0, 2, 2), 50)
    print("2D 1025^2 7-level V(2,2) %s: %.3f ms  %.2f GLUPS" % (name, t * 1e3, lups(1025, 7, 2) / t / 1e9))
    mg.use_graph = True
    t = timed(lambda: mg.VCycle(0, 2, 2), 50)
    print("2D 1025^2 7-level V(2,2) %s, HIP graph replay: %.3f ms  %.2f GLUPS" % (name, t * 1e3, lups(1025, 7, 2) / t / 1e9))
    mg.close()
mg = P.MultiGrid3D(ctx, [129] * 3, R3, np.float32)
t = timed(lambda: mg.FullMultiGridVCycle(0, 2, 3000, 3000), 1)
print("3D 129^3 FMG(2,3000,3000) f32 (thesis parameters; reference GPU: 39.1 s on GTX 550 Ti): %.2f s" % t)
mg.use_graph = True
t = timed(lambda: mg.FullMultiGridVCycle(0, 2, 3000, 3000), 1)
print("3D 129^3 FMG(2,3000,3000) f32, HIP graph replay: %.2f s" % t)
