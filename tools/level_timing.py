#!/usr/bin/python3
"""Where a V(2,2) cycle spends its time, WITHOUT a profiler (rocprofv3 inflates short kernels): the cycle started at level g
is timed with HIP events for every g; level g's own share is the difference to the cycle started at g+1.  Also times the
finest level's operators one by one.
    python tools/level_timing.py [n] [f64|f32]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
ctx = P.Context(0)
for k, v in [a.split("=") for a in os.environ.get("MGX_PARAMS", "").split(",") if a]:
    ctx.set_param(k, int(v))
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype)
e0, e1 = ctx.event(), ctx.event()
REPS = 20


def timed(fn, reset=None):
    tot = 0.0
    for i in range(REPS + 2):
        if reset:
            reset()
        ctx.sync()
        ctx.record(e0)
        fn()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1)
        if i >= 2:
            tot += ms
    return tot / REPS


ts = []
for g in range(mg.numGrids):
    ts.append(timed(lambda: mg.VCycle(g, 2, 2), lambda: mg.setToValue_v(g, 0.0, True)))
ts.append(0.0)
print("V(2,2) %d^3 %s, %d levels: %.4f ms" % (n, np.dtype(dtype).name, mg.numGrids, ts[0]))
for g in range(mg.numGrids):
    print("  level %d (%4d^3): cycle from here %.4f ms, own share %.4f ms" % (g, mg.size(g)[0], ts[g], ts[g] - ts[g + 1]))
print("finest level operators:")
print("  Relax(2) [4 colour passes]      %.4f ms" % timed(lambda: mg.Relax(0, 2)))
g0, g1 = mg.grid(0), mg.grid(1)
import ctypes as C  # noqa: E402
sfx = "f64" if dtype == np.float64 else "f32"
ct = C.c_double if dtype == np.float64 else C.c_float
h = (ct * 3)(g0.h_x, g0.h_y, g0.h_z)
n0, n1 = (C.c_int * 3)(*g0.sizeXYZ), (C.c_int * 3)(*g1.sizeXYZ)
rr = getattr(P.lib, "mgx3dxs_residual_restrict_keep_rim_" + sfx)
ic = getattr(P.lib, "mgx3dxs_interpolate_correct_colour_" + sfx)
print("  residual+restrict               %.4f ms" % timed(lambda: P.check(rr(ctx._h, C.c_void_p(g0.d_v), C.c_void_p(g0.d_f), n0, h, C.c_int(0), C.c_void_p(g1.d_f), n1))))
srr = getattr(P.lib, "mgx3dxs_smooth_residual_restrict_" + sfx)
icr = getattr(P.lib, "mgx3dxs_interpolate_correct_relax_" + sfx)
print("  the way down in one call: Relax(2) + residual + restrict   %.4f ms  [%s]" % (
    timed(lambda: P.check(srr(ctx._h, C.c_void_p(g0.d_v), C.c_void_p(g0.d_f), n0, h, C.c_int(2), C.c_int(0), C.c_int(0), C.c_int(0),
                              C.c_void_p(g1.d_f), n1, C.c_int(1)))), ctx.last_rr_kernel() or "separate launches"))
print("  the way up in one call: interpolate + correct + Relax(2)   %.4f ms" % timed(
    lambda: P.check(icr(ctx._h, C.c_void_p(g0.d_v), C.c_void_p(g0.d_f), n0, h, C.c_void_p(g1.d_v), n1, C.c_int(2)))))
print("  interpolate+correct (black)     %.4f ms" % timed(lambda: P.check(ic(ctx._h, C.c_void_p(g0.d_v), n0, C.c_void_p(g1.d_v), n1, C.c_int(1)))))
print("  zero fill of the coarse v       %.4f ms" % timed(lambda: mg.setToValue_v(1, 0.0, True)))
