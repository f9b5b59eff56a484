"""time residual+restrict (keep_rim entry, x-split layout) of the finest level of an n^3 hierarchy:  rr_time.py n f64|f32"""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P
from pde_multigrid_amd._lib import lib, check
n = int(sys.argv[1]); dtype = np.float32 if sys.argv[2] == "f32" else np.float64
sfx = "f32" if dtype == np.float32 else "f64"
ct = C.c_float if dtype == np.float32 else C.c_double
ctx = P.Context(0)
for k, v in [a.split("=") for a in os.environ.get("MGX_PARAMS", "").split(",") if a]:
    ctx.set_param(k, int(v))
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype, nlevels=2)
mg.Relax(0, 1)
g0, g1 = mg.grid(0), mg.grid(1)
fn = getattr(lib, "mgx3dxs_residual_restrict_keep_rim_" + sfx)
h = (ct * 3)(g0.h_x, g0.h_y, g0.h_z)
ip = lambda a: (C.c_int * 3)(*a)
def run():
    check(fn(ctx._h, C.c_void_p(g0.d_v), C.c_void_p(g0.d_f), ip(g0.sizeXYZ), h, C.c_int(0), C.c_void_p(g1.d_f), ip(g1.sizeXYZ)))
e0, e1 = ctx.event(), ctx.event()
run(); run(); ctx.sync(); ctx.record(e0)
for _ in range(20): run()
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / 20
b = ((n - 0) ** 3 * 2 + ((n + 1) // 2) ** 3) * np.dtype(dtype).itemsize
print("%d %s residual+restrict %.1f us  %.0f GB/s algorithmic  frac %.3f  [%s]" % (n, sfx, ms * 1e3, b / ms / 1e6, b / ms / 1e6 / 8000, os.environ.get("MGX_PARAMS", "")))
