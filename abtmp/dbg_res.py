import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import oracle as O
import pde_multigrid_amd as P
RG = [-1, 1, 0, 2, 0.5, 3]
ctx = P.Context(0)
ctx.set_param("relax3d.resident", 1)
ctx.set_param("relax3d.resident_tile", 0)
ctx.set_param("relax3d.resident_min", 1)
sizes = [(33, 33, 33), (65, 65, 65), (129, 129, 129), (129, 65, 33), (65, 129, 17), (33, 9, 129), (129, 17, 9)]
for rep in range(3):
  for nc in (2, 3):
    for n3 in sizes:
        for dtype in (np.float64, np.float32):
            r = np.random.default_rng(n3[0] + nc)
            shape = tuple(reversed(n3))
            v, f = r.uniform(-1, 1, shape).astype(dtype), r.uniform(-1, 1, shape).astype(dtype)
            got = P.ops3dxs.relax(ctx, v, f, n3, RG, nc)
            ctx.sync()
            want = O.relax3d(n3, RG, v, f, nc, dtype=dtype)
            U = np.uint64 if dtype == np.float64 else np.uint32
            bad = np.argwhere(got.view(U) != want.view(U))
            print(n3, nc, np.dtype(dtype).name, ctx.last_relax_kernel(), "mismatches", len(bad), flush=True)
            if len(bad):
                zs, ys, xs = bad[:, 0], bad[:, 1], bad[:, 2]
                print("  z:", sorted(set(zs.tolist())), "y:", sorted(set(ys.tolist())), "x:", sorted(set(xs.tolist()))[:40], "...")
