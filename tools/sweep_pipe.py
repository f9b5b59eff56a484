#!/usr/bin/python3
"""Sweep workgroup shape and z-chunk count of relax3d_xs_pipe_kernel ("relax3d.lds" = 1000 + 100*WX + 10*WY + R).
    python tools/sweep_pipe.py --n=513 [--dtype=f32] [--planes=NZ]   (NZ: slab height, default n)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n, dt = 513, np.float64
for a in sys.argv[1:]:
    if a.startswith("--n="):
        n = int(a[4:])
    if a == "--dtype=f32":
        dt = np.float32
ctx = P.Context(0)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dt, nlevels=2)
e0, e1 = ctx.event(), ctx.event()
M1 = (n + 1) // 2 - 1
codes = [c for c in (1442, 1422, 1444, 1424, 1242, 1282, 1244, 1224, 1144, 1184, 1824, 1814, 1414)
         if M1 >= 64 * ((c % 1000) // 100) and n - 2 >= ((c // 10) % 10) * (c % 10)]
combos = [(0, 0)]
for c in codes:
    for nch in (1, 2, 3, 4, 6, 8, 12, 16, 32):
        z = -(-(n - 2) // nch)
        combos.append((c, z))
res = {k: [] for k in combos}
for rnd in range(3):
    for (c, z) in combos:
        ctx.set_param("relax3d.lds", c)
        ctx.set_param("relax3d.zchunk", z)
        ctx.sync(); ctx.record(e0)
        for _ in range(4):
            mg.Relax(0, 2)
        ctx.record(e1)
        t = ctx.elapsed_ms(e0, e1) / 8
        if rnd:
            res[(c, z)].append(t)
out = sorted((np.median(v), k) for k, v in res.items())
for t, (c, z) in out[:25]:
    print("n=%d %s lds=%4d zchunk=%4d  %.4f ms/sweep  %.1f GLUPS" % (n, np.dtype(dt).name, c, z, t, (n - 2) ** 3 / t / 1e6))
print("baseline", [("%.4f" % t) for t, k in out if k == (0, 0)])
