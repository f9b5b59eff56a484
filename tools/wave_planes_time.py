#!/usr/bin/python3
"""Relax(k) on one level with the colour passes time-skewed over z-slabs of B planes ("relax3d.wave_planes"): does a slab that fits
the 256 MiB Infinity Cache make the passes after the first faster?
    python3 tools/wave_planes_time.py [n=513] [f64|f32] [sweeps=2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ctx = P.Context(0)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype)
e0, e1 = ctx.event(), ctx.event()


def timed(reps=12):
    ts = []
    for i in range(reps + 3):
        ctx.sync()
        ctx.record(e0)
        mg.Relax(0, k)
        ctx.record(e1)
        ts.append(ctx.elapsed_ms(e0, e1))
    ts = sorted(ts[3:])
    return ts[len(ts) // 2], ts[0]


for B in (0, 8, 12, 16, 24, 32, 48, 64, 96, 128, 0):
    ctx.set_param("relax3d.wave_planes", B)
    med, mn = timed()
    print("n=%d %s Relax(%d) wave_planes=%3d: median %.4f ms (min %.4f)  [%s]" % (n, np.dtype(dtype).name, k, B, med, mn, ctx.last_relax_kernel()), flush=True)
