#!/usr/bin/python3
"""One colour pass over a run of planes of a z-slab (mgx3dxs_relax_colour_slab / _slab2): microseconds per launch by plane size, run
length and kernel choice ("relax3d.lds" = -1: automatic, 0: relax3d_xs_kernel).  The edge passes of the communication-avoiding slab
schedule are runs of 4 ... 11 planes, the interior of a thin slab 20 ... 60.
    python tools/slab_pass_time.py [fp64|fp32]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

sfx = "f32" if len(sys.argv) > 1 and sys.argv[1] == "fp32" else "f64"
ct = C.c_float if sfx == "f32" else C.c_double
wb = 4 if sfx == "f32" else 8
ctx = P.Context(0)
fn = getattr(P.lib, "mgx3dxs_relax_colour_slab_" + sfx)
fn2 = getattr(P.lib, "mgx3dxs_relax_colour_slab2_" + sfx)
pe = getattr(P.lib, "mgx3dxs_plane_elems_" + sfx)
pe.restype = C.c_size_t
e0, e1 = ctx.event(), ctx.event()
for n in (1025, 513, 257):
    pl = pe(n, n)
    nzmax = 80
    v = ctx.malloc(pl * nzmax * wb)
    f = ctx.malloc(pl * nzmax * wb)
    P.check(P.lib.mgx_memset_zero(ctx._h, v, C.c_size_t(pl * nzmax * wb)))
    P.check(P.lib.mgx_memset_zero(ctx._h, f, C.c_size_t(pl * nzmax * wb)))
    h = (ct * 3)(1.0 / (n - 1), 1.0 / (n - 1), 1.0 / (n - 1))
    for nz in (2, 4, 5, 7, 9, 11, 16, 21, 32, 38, 64, 70):
        row = []
        for lds in (-1, 0):
            ctx.set_param("relax3d.lds", lds)
            for rep in range(2):
                ctx.sync()
                ctx.record(e0)
                for k in range(20):
                    P.check(fn(ctx._h, v, f, n, n, h, k & 1, 1, 1 + nz, 0))
                ctx.record(e1)
                ms = ctx.elapsed_ms(e0, e1)
            row.append((ms / 20 * 1e3, ctx.last_relax_kernel()))
        # two runs of nz planes in one launch
        ctx.set_param("relax3d.lds", -1)
        two = None
        if 2 * nz + 4 <= nzmax:
            for rep in range(2):
                ctx.sync()
                ctx.record(e0)
                for k in range(20):
                    P.check(fn2(ctx._h, v, f, n, n, h, k & 1, 1, 1 + nz, 3 + nz, 3 + 2 * nz, 0))
                ctx.record(e1)
                ms = ctx.elapsed_ms(e0, e1)
            two = (ms / 20 * 1e3, ctx.last_relax_kernel())
        ideal = 3 * wb * (n - 2) ** 2 * nz / 2 / 5.6e12 * 1e6
        print("%5d^2 x %2d planes: auto %7.1f us (%s)   small kernel %7.1f us   two runs in one call %s   [streaming %5.1f us]"
              % (n, nz, row[0][0], row[0][1][:44], row[1][0], ("%7.1f us (%s)" % (two[0], two[1][:30])) if two else "-", ideal), flush=True)
    ctx.free(v)
    ctx.free(f)
ctx.close()
