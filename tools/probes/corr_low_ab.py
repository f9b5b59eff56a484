import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import oracle as O
import pde_multigrid_amd as P
ctx = P.Context(0)
# correctness at a small wide size
n3 = (513, 129, 33); rg = [-1, 1, 0, 2, 0.5, 3]
r = np.random.default_rng(3)
v = r.uniform(-1, 1, O.shape(n3)); f = r.uniform(-1, 1, O.shape(n3)); c = r.uniform(-1, 1, O.shape(P.coarse_size(n3)))
want = O.relax3d(n3, rg, O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3)), c, dtype=np.float64), dtype=np.float64), f, 1, dtype=np.float64)
for low in (0, 1):
    ctx.set_param("relax3d.corr_low", low)
    got = P.ops3dxs.interpolate_correct_relax(ctx, v, f, n3, rg, c, 1)
    print("low", low, ctx.last_corr_kernel(), "bit-exact:", bool((got.view(np.uint64) == want.view(np.uint64)).all()), flush=True)
# timing at 513^3: the way up of the finest level
mg = P.MultiGrid3D(ctx, [513] * 3, [0, 1, 0, 1, 0, 1], np.float64)
for _ in range(60): mg.VCycle(0, 2, 2)
e0, e1 = ctx.event(), ctx.event()
for rep in range(3):
    for low in (0, 1):
        ctx.set_param("relax3d.corr_low", low)
        for _ in range(5): mg.VCycle(0, 2, 2)
        ctx.sync(); ctx.record(e0)
        for _ in range(30): mg.VCycle(0, 2, 2)
        ctx.record(e1); ctx.sync()
        print("corr_low", low, "cycle %.4f ms" % (ctx.elapsed_ms(e0, e1) / 30), ctx.last_corr_kernel(), flush=True)
