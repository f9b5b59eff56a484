#!/usr/bin/python3
"""TIMING ONLY (diagnostic library, wrong results): the unrolled plain colour pass with the loads and stores of a colour-contiguous
layout (every pass streams three contiguous half-planes) against the x-split layout's (one 2112-byte half of every 4224-byte row).
    python3 tools/cs_pattern_time.py [n=513] [f64|f32] [relax3d.lds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MGX_LIB_PATH", os.path.join(ROOT, "pde_multigrid_amd", "lib", "libmgx_diag.so"))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
ctx = P.Context(0)
if dtype == np.float32:
    ctx.set_param("relax3d.v2", 0)
    ctx.set_param("relax3d.unroll", 15)
if len(sys.argv) > 3:
    ctx.set_param("relax3d.lds", int(sys.argv[3]))  # e.g. 1442: tiles of 256 pairs x 8 rows (the whole half-row per workgroup)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype)
e0, e1 = ctx.event(), ctx.event()


def timed(reps=15):
    ts = []
    for i in range(reps + 3):
        ctx.sync()
        ctx.record(e0)
        mg.Relax(0, 2)
        ctx.record(e1)
        ts.append(ctx.elapsed_ms(e0, e1))
    ts = sorted(ts[3:])
    return ts[len(ts) // 2], ts[0]


for name, abl in (("x-split pattern", 0), ("colour-contiguous pattern", 77), ("x-split pattern", 0), ("colour-contiguous pattern", 77)):
    ctx.set_param("relax3d.ablate", abl)
    med, mn = timed()
    print("n=%d %s Relax(2) = 4 colour passes, %-26s median %.4f ms (min %.4f) = %.1f us per pass  [%s]" % (
        n, np.dtype(dtype).name, name + ":", med, mn, med * 250, ctx.last_relax_kernel()), flush=True)
