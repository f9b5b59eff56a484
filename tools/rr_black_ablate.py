#!/usr/bin/python3
"""Ablation timing of relax_rr3d_xs_kernel on the DIAGNOSTIC library (make -C pde_multigrid_amd/csrc diag): parts of an
iteration switched off one at a time (results are WRONG; timing only).
    python3 tools/rr_black_ablate.py [n=513] [f64|f32]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MGX_LIB_PATH", os.path.join(ROOT, "pde_multigrid_amd", "lib", "libmgx_diag.so"))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P  # noqa: E402
from pde_multigrid_amd.multigrid import _ip, _rp, coarse_size, grid_spacing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
sfx, ct = ("f32", C.c_float) if dtype == np.float32 else ("f64", C.c_double)
n3, cn = [n] * 3, coarse_size([n] * 3)
ctx = P.Context(0)
elems = getattr(P.lib, "mgx3dxs_elems_" + sfx)
elems.restype = C.c_size_t
nv, nc = elems(_ip(n3)), elems(_ip(cn))
r = np.random.default_rng(1)
pv = ctx.to_device(r.uniform(-1, 1, nv).astype(dtype))
pf = ctx.to_device(r.uniform(-1, 1, nv).astype(dtype))
pc = ctx.to_device(np.zeros(nc, dtype))
h = _rp(grid_spacing(n3, [0, 1, 0, 1, 0, 1], dtype), ct)
fn = getattr(P.lib, "mgx3dxs_smooth_residual_restrict_" + sfx)
ev0, ev1 = ctx.event(), ctx.event()


def run(v1, reps):
    ts = []
    for _ in range(reps):
        ctx.record(ev0)
        P.check(fn(ctx._h, pv, pf, _ip(n3), h, C.c_int(v1), C.c_int(0), C.c_int(0), C.c_int(P.REF_COMPAT), pc, _ip(cn), C.c_int(1)))
        ctx.record(ev1)
        ctx.sync()
        ts.append(ctx.elapsed_ms(ev0, ev1))
    ts.sort()
    return ts[len(ts) // 2]


ctx.set_param("rr3d.black", 0)
run(1, 3)
base = run(1, 11)  # red pass + black pass + residual+restrict
ctx.set_param("rr3d.black", 1)
names = {0: "everything", 1: "no loads", 2: "no stores of v", 4: "no relax arithmetic", 8: "no residual arithmetic", 16: "no barrier",
         32: "no sub-sums / coarse rows", 12: "no arithmetic", 44: "no arithmetic, no sub-sums", 3: "no loads, no stores",
         47: "barrier and LDS traffic only", 63: "nothing but the loop"}
red = None
for abl in (0, 1, 2, 3, 4, 8, 12, 16, 32, 44, 47, 63):
    if abl:
        ctx.set_param("rr3d.black_abl", abl)
    run(1, 3)
    t = run(1, 11)
    print("n=%d %s  abl=%2d %-34s red pass + fused launch %.4f ms   (separate launches %.4f ms)  [%s]" % (
        n, np.dtype(dtype).name, abl, names[abl] + ":", t, base, ctx.last_rr_kernel()), flush=True)
ctx.close()
