#!/usr/bin/python3
"""The reference's own published workloads (thesis ch. 4, BASELINE.md section 1: whole-program wall times on a Pentium
E5400 / GeForce GTX 550 Ti) on one MI355X: hierarchy construction + RHS initialisation + full-multigrid solve + download of
the solution, fp32 like the reference.  Parameters: thesis pp. 69-72 and the reference's main programs
(NOCUDA_TESI/*/ *Solver.cpp).

    python3 tools/thesis_workloads.py [quick]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
ctx = P.Context(0)
ctx.sync()


def wall(f):
    ctx.sync()
    t0 = time.perf_counter()
    r = f()
    ctx.sync()
    return time.perf_counter() - t0, r


def run3(n):
    def go():
        mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float32)
        mg.FullMultiGridVCycle(0, 2, 3000, 3000)
        v = mg.download_v(0)
        mg.close()
        return v
    return wall(go)


def run2(n, rng):
    def go():
        mg = P.MultiGrid2D(ctx, [n, n], rng, [-1.0, -2.0, 0.0, -3.0], 2, np.float32)
        mg.FullMultiGridVCycle(0, 2, 500, 500)
        v = mg.download_v(0)
        mg.close()
        return v
    return wall(go)


def run1(n):
    def go():
        mg = P.MultiGrid1D(n, [0, 1], np.float32)
        mg.FullMultiGridVCycle(0, 2, 1000, 1000)
        return None
    t0 = time.perf_counter()
    go()
    return time.perf_counter() - t0, None


PUB3 = {9: (1.6, 1.1), 17: (2.0, 3.3), 33: (2.7, 23.0), 65: (6.7, 213.4), 129: (39.1, None), 257: (295.2, None)}
PUB2 = {65: (2.1, 2.1), 129: (2.3, 3.4), 257: (2.4, 9.3), 513: (2.8, 32.9), 1025: (3.8, 127.8), 2049: (7.6, 508.6), 4097: (21.4, None)}
PUB1 = {257: (1.1, 0.5), 513: (1.4, 1.0), 1025: (1.7, 2.0), 2049: (2.0, 4.2), 4097: (2.3, 7.8), 8193: (2.7, 15.6)}


def line(what, n, t, pub):
    g, c = pub
    print("%-46s n = %5d: %8.3f s   (thesis: GPU %s s, CPU %s s)" % (what, n, t, g, "-" if c is None else c), flush=True)


run3(9)  # warm-up (module load)
for n in ([33, 129] if quick else [9, 17, 33, 65, 129, 257]):
    t, _ = run3(n)
    line("3D Poisson FMG(2,3000,3000) [0,1]^3 fp32", n, t, PUB3[n])
for n in ([129, 1025] if quick else [65, 129, 257, 513, 1025, 2049, 4097]):
    t, _ = run2(n, [0, 20, 0, 20])
    line("2D Lyapunov FMG(2,500,500) [0,20]^2 fp32", n, t, PUB2[n])
for n in ([257] if quick else [257, 513, 1025, 2049, 4097, 8193]):
    t, _ = run1(n)
    line("1D ODE FMG(2,1000,1000) [0,1] fp32 (host C)", n, t, PUB1[n])
