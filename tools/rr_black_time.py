#!/usr/bin/python3
"""Time the way down on the finest level: Relax(v1) + residual + restrict as separate launches against the form with the
last black pass inside the residual+restrict launch (mgx3dxs_smooth_residual_restrict, csrc/mgx_relax_rr3d.hip).
    python3 tools/rr_black_time.py [n=513] [f64|f32] [v1=2] [name=value,...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402
from pde_multigrid_amd.multigrid import _ip, _rp, coarse_size, grid_spacing  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
v1 = int(sys.argv[3]) if len(sys.argv) > 3 else 2
extra = sys.argv[4] if len(sys.argv) > 4 else ""
sfx, ct = ("f32", C.c_float) if dtype == np.float32 else ("f64", C.c_double)
n3, cn = [n] * 3, coarse_size([n] * 3)
ctx = P.Context(0)
for kv in filter(None, extra.split(",")):
    k, val = kv.split("=")
    ctx.set_param(k, int(val))
elems = getattr(P.lib, "mgx3dxs_elems_" + sfx)
elems.restype = C.c_size_t
isz = np.dtype(dtype).itemsize
nv, nc = elems(_ip(n3)), elems(_ip(cn))
r = np.random.default_rng(1)
pv = ctx.to_device(r.uniform(-1, 1, nv).astype(dtype))
pf = ctx.to_device(r.uniform(-1, 1, nv).astype(dtype))
pc = ctx.to_device(np.zeros(nc, dtype))
h = _rp(grid_spacing(n3, [0, 1, 0, 1, 0, 1], dtype), ct)
fn = getattr(P.lib, "mgx3dxs_smooth_residual_restrict_" + sfx)


def run(reps):
    ev0, ev1 = ctx.event(), ctx.event()
    ts = []
    for _ in range(reps):
        ctx.record(ev0)
        P.check(fn(ctx._h, pv, pf, _ip(n3), h, C.c_int(v1), C.c_int(0), C.c_int(0), C.c_int(P.REF_COMPAT), pc, _ip(cn), C.c_int(1)))
        ctx.record(ev1)
        ctx.sync()
        ts.append(ctx.elapsed_ms(ev0, ev1))
    return ts


ON = 2 if dtype == np.float32 else 1  # fp32 is not taken by default: force it
cases = [("separate launches", {"rr3d.black": 0}), ("black pass inside, 16 waves", {"rr3d.black": ON, "rr3d.black_waves": 16}),
         ("black pass inside, 12 waves", {"rr3d.black": ON, "rr3d.black_waves": 12}),
         ("black pass inside, 8 waves x 2 per CU", {"rr3d.black": ON, "rr3d.black_waves": 8}), ("separate launches", {"rr3d.black": 0})]
for name, params in cases:
    for k, val in params.items():
        ctx.set_param(k, val)
    run(3)
    ts = sorted(run(15))
    print("n=%d %s Relax(%d) + residual + restrict, %-30s median %.4f ms (min %.4f)  [%s]" % (
        n, np.dtype(dtype).name, v1, name + ":", ts[len(ts) // 2], ts[0], ctx.last_rr_kernel() or "-"), flush=True)
ctx.close()
