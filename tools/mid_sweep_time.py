#!/usr/bin/python3
"""Time one red+black sweep launch of sweep3d_xs_mid_kernel (levels of 33 ... 65-point rows), plain and from-zero form.
    python3 tools/mid_sweep_time.py [f64|f32]          (MGX_LIB_PATH picks the library: A/B on one box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

dtype = np.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else np.float64
ctx = P.Context(0)
ctx.set_param("relax3d.resident", 0)
for n in (33, 65):
    n3 = (n, n, n)
    mg = P.MultiGrid3D(ctx, n3, [0, 1, 0, 1, 0, 1], dtype, nlevels=1)
    e0, e1 = ctx.event(), ctx.event()
    for sweeps in (2, 20):
        ts = []
        for _ in range(30):
            ctx.sync()
            ctx.record(e0)
            mg.Relax(0, sweeps)
            ctx.record(e1)
            ts.append(ctx.elapsed_ms(e0, e1) * 1e3)
        ts.sort()
        print("%s %d^3 Relax(%d): median %.2f us, min %.2f  -> %.2f us per sweep  [%s]" % (
            np.dtype(dtype).name, n, sweeps, ts[15], ts[0], ts[15] / sweeps, ctx.last_relax_kernel()), flush=True)
    mg.close()
ctx.close()
