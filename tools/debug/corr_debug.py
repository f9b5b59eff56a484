import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import pde_multigrid_amd as P
ctx = P.Context(0)
n3 = tuple(int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (257, 129, 33)
dtype = np.float64
rg = [-1, 1, 0, 2, 0.5, 3]
rng = np.random.default_rng(sum(n3))
cn = P.coarse_size(n3)
v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
corrected = O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3), dtype), c, dtype=dtype), dtype=dtype)
for fuse in (1, 0):
    ctx.set_param("relax3d.corr_fuse", fuse)
    for zchunk in (0, 3):
        ctx.set_param("relax3d.zchunk", zchunk)
        for k in (1,):
            want = O.relax3d(n3, rg, corrected, f, k, dtype=dtype)
            got = P.ops3dxs.interpolate_correct_relax(ctx, v, f, n3, rg, c, k)
            bad = np.argwhere(got.view(np.uint64) != want.view(np.uint64))
            print("fuse", fuse, "zchunk", zchunk, "k", k, "mismatches", len(bad), "of", got.size)
            if len(bad):
                z, y, x = bad[:, 0], bad[:, 1], bad[:, 2]
                print("  colour (x+y+z)&1:", np.bincount((x + y + z) & 1, minlength=2))
                print("  z values:", np.unique(z)[:40])
                print("  y mod 16:", np.bincount(y % 16, minlength=16))
                print("  (x>>1) mod 128 hist (nonzero):", {int(a): int(b) for a, b in zip(*np.unique((x >> 1) % 128, return_counts=True))} if len(bad) < 200000 else "many")
                print("  first:", bad[:10].tolist())
                # compare against the uncorrected relax (correction missing) and double-corrected
                w0 = O.relax3d(n3, rg, v, f, k, dtype=dtype)
                print("  equal to relax WITHOUT correction at bad points:", int((got[z, y, x] == w0[z, y, x]).sum()))
