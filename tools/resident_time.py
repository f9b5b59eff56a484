#!/usr/bin/python3
"""Relax(ncycles) on the cache-resident levels: all passes in one launch (relax3d_xs_resident_kernel) against one launch per
colour pass / per sweep.
    python3 tools/resident_time.py [f64|f32]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402
from pde_multigrid_amd.multigrid import _ip, _rp, grid_spacing  # noqa: E402

dtype = np.float32 if len(sys.argv) > 1 and sys.argv[1] == "f32" else np.float64
sfx, ct = ("f32", C.c_float) if dtype == np.float32 else ("f64", C.c_double)
ctx = P.Context(0)
elems = getattr(P.lib, "mgx3dxs_elems_" + sfx)
elems.restype = C.c_size_t
relax = getattr(P.lib, "mgx3dxs_relax_" + sfx)
e0, e1 = ctx.event(), ctx.event()
for n in (33, 65, 129):
    n3 = [n] * 3
    ne = elems(_ip(n3))
    r = np.random.default_rng(n)
    pv = ctx.to_device(r.uniform(-1, 1, ne).astype(dtype))
    pf = ctx.to_device(r.uniform(-1, 1, ne).astype(dtype))
    h = _rp(grid_spacing(n3, [0, 1, 0, 1, 0, 1], dtype), ct)
    for ncycles in (2, 3, 10, 100, 3000):
        res = {}
        for resident in (0, 2, 8, 1):
            ctx.set_param("relax3d.resident", 1 if resident == 8 else resident)
            ctx.set_param("relax3d.resident_tile", 8 if resident == 8 else 0)
            ctx.set_param("relax3d.resident_min", 1)
            ts = []
            for i in range(7 if ncycles < 1000 else 3):
                ctx.record(e0)
                P.check(relax(ctx._h, pv, pf, _ip(n3), h, C.c_int(ncycles)))
                ctx.record(e1)
                ctx.sync()
                ts.append(ctx.elapsed_ms(e0, e1))
            res[resident] = (sorted(ts)[len(ts) // 2], ctx.last_relax_kernel())
        print("%3d^3 %s Relax(%4d): per-pass launches %9.4f ms (%.2f us per pass) [%s]   one launch, exchange per pass %9.4f ms (%.2f us per pass)"
              "   exchange per sweep, tiles of 8: %9.4f ms (%.2f us per pass)   exchange per sweep %9.4f ms (%.2f us per pass) [%s]" % (
                  n, np.dtype(dtype).name, ncycles, res[0][0], res[0][0] * 1e3 / (2 * ncycles), res[0][1].split("<")[0], res[2][0],
                  res[2][0] * 1e3 / (2 * ncycles), res[8][0], res[8][0] * 1e3 / (2 * ncycles), res[1][0], res[1][0] * 1e3 / (2 * ncycles),
                  res[1][1]), flush=True)
    ctx.free(pv)
    ctx.free(pf)
ctx.close()
