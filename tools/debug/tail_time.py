import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P
ctx = P.Context(0)
for dim, n, dtype in ((3, 17, np.float64), (3, 9, np.float64), (3, 17, np.float32), (2, 33, np.float64), (2, 17, np.float64)):
    if dim == 3:
        mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype)
    else:
        mg = P.MultiGrid2D(ctx, [n] * 2, [0, 1, 0, 1], [-1.0, -2.0, 0.0, -3.0], 2, dtype)
    e0, e1 = ctx.event(), ctx.event()
    for v in (0, 1, 2, 4, 8):
        if dim == 2 and v > 4: continue
        mg.VCycle(0, v, v); ctx.sync(); ctx.record(e0)
        for _ in range(50): mg.VCycle(0, v, v)
        ctx.record(e1)
        print("%dD %d %s V(%d,%d): %.2f us" % (dim, n, np.dtype(dtype).name, v, v, ctx.elapsed_ms(e0, e1) / 50 * 1e3), flush=True)
    mg.close()
