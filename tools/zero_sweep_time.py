import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P
from pde_multigrid_amd.multigrid import _ip, _rp, grid_spacing
ctx = P.Context(0)
for n, dt, sfx, ct in ((257, np.float64, "f64", C.c_double), (513, np.float64, "f64", C.c_double), (257, np.float32, "f32", C.c_float)):
    n3 = [n]*3
    el = getattr(P.lib, "mgx3dxs_elems_" + sfx); el.restype = C.c_size_t
    ne = el(_ip(n3))
    pv = ctx.to_device(np.zeros(ne, dt)); pf = ctx.to_device(np.random.default_rng(0).uniform(-1, 1, ne).astype(dt))
    h = _rp(grid_spacing(n3, [0,1,0,1,0,1], dt), ct)
    fn = getattr(P.lib, "mgx3dxs_relax_from_zero_" + sfx)
    e0, e1 = ctx.event(), ctx.event()
    for zs in (1, 0, 1, 0):
        ctx.set_param("relax3d.zero_sweep", zs)
        ts = []
        for i in range(12):
            ctx.record(e0); P.check(fn(ctx._h, pv, pf, _ip(n3), h, C.c_int(1), C.c_int(1))); ctx.record(e1); ctx.sync()
            ts.append(ctx.elapsed_ms(e0, e1))
        print(n, sfx, "zero_sweep", zs, "first sweep from zero: median %.4f ms min %.4f [%s]" % (sorted(ts)[6], min(ts), ctx.last_relax_kernel()), flush=True)
    ctx.free(pv); ctx.free(pf)
