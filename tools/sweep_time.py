#!/usr/bin/python3
"""Relax(2) on the finest level of an n^3 hierarchy: one launch per red+black sweep (mgx_sweep3d.hip) against one launch
per colour, HIP-event timed, same box, same process.
    python tools/sweep_time.py [n] [f64|f32] [lead,...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
leads = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
ctx = P.Context(0)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype, nlevels=2)
e0, e1 = ctx.event(), ctx.event()
REPS = 20


def timed(fn):
    ts = []
    for i in range(REPS + 3):
        ctx.sync()
        ctx.record(e0)
        fn()
        ctx.record(e1)
        ts.append(ctx.elapsed_ms(e0, e1))
    ts = sorted(ts[3:])
    return ts[len(ts) // 2], ts[0]


pts = float(n - 2) ** 3
for fused, lead, ilv in [(0, 0, 0)] + [(1, l, i) for l in leads for i in (0, 1)] + [(0, 0, 0)]:
    ctx.set_param("relax3d.fused", fused)
    ctx.set_param("relax3d.fused_lead", lead)
    ctx.set_param("relax3d.fused_ilv", ilv)
    med, best = timed(lambda: mg.Relax(0, 2))
    ctx.sync()
    print("n=%d %s fused=%d lead=%d ilv=%d: Relax(2) median %.4f ms (min %.4f) = %.4f ms per sweep, %.1f GLUPS  [%s]" % (
        n, np.dtype(dtype).name, fused, lead, ilv, med, best, med / 2, 2 * pts / med / 1e6, ctx.last_relax_kernel()), flush=True)
mg.close()
ctx.close()
