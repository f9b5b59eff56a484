"""Does the smoother's speed depend on WHERE its arrays were allocated?  One process, the 513^3 hierarchy built several times (with
dummy allocations of varying size in between to shift the addresses), the plain colour pass timed each time in the steady state."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pde_multigrid_amd as P
ctx = P.Context(0)
for kv in filter(None, os.environ.get("MGX_PARAMS", "").split(",")):
    ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))
e0, e1 = ctx.event(), ctx.event()
warm = P.MultiGrid3D(ctx, [513] * 3, [0, 1, 0, 1, 0, 1], np.float64, nlevels=1)
for _ in range(60): warm.Relax(0, 2)
keep = []
for trial in range(8):
    if trial % 2 == 1:
        keep.append(ctx.malloc(int((trial * 37 + 5) * (1 << 20) + trial * 4096)))  # shift the next allocations
    mg = P.MultiGrid3D(ctx, [513] * 3, [0, 1, 0, 1, 0, 1], np.float64, nlevels=1)
    g = mg.grid(0)
    ts = []
    for rep in range(3):
        for _ in range(10): mg.Relax(0, 2)
        ctx.sync(); ctx.record(e0)
        for _ in range(20): mg.Relax(0, 2)
        ctx.record(e1); ctx.sync()
        ts.append(ctx.elapsed_ms(e0, e1) / 80 * 1e3)
    print("trial %d  d_v %#x  d_f %#x  (f - v) mod 2^21 = %#x   colour pass %.2f / %.2f / %.2f us" % (
        trial, g.d_v, g.d_f, (g.d_f - g.d_v) % (1 << 21), ts[0], ts[1], ts[2]), flush=True)
    mg.close()
