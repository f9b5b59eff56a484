import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
import pde_multigrid_amd as P
ctx = P.Context(0)
A = [-1.0, -2.0, 0.0, -3.0]
for n, box, dtype, nlev, v1, v2, reps in (([65, 65], [0.0, 20.0, -1.0, 19.0], np.float64, 0, 1, 3, 2), ([129, 129], [0.5, 20.5, -1.0, 0.5], np.float32, 0, 2, 1, 2)):
    rng = np.random.default_rng(3)
    v = rng.uniform(-1, 1, O.shape(n)).astype(dtype); f = rng.uniform(-1, 1, O.shape(n)).astype(dtype)
    want = O.cycle2d(n, box, A, 2, nlevels=nlev, mode=0, v1=v1, v2=v2, reps=reps, v=v, f=f, dtype=dtype)
    for fuse in (2, 1, 0):
        mg = P.MultiGrid2D(ctx, n, box, A, 2, dtype, nlevels=nlev, fuse=fuse)
        mg.upload_v(0, v); mg.upload_f(0, f)
        for _ in range(reps): mg.VCycle(0, v1, v2)
        got = mg.download_v(0); mg.close()
        neq = got.view(np.uint8).reshape(got.shape + (-1,)) != want.view(np.uint8).reshape(want.shape + (-1,))
        bad = neq.any(-1)
        print(n, box, np.dtype(dtype).name, "fuse", fuse, "bit mismatches", int(bad.sum()), "nan got/want", int(np.isnan(got).sum()), int(np.isnan(want).sum()),
              "inf", int(np.isinf(got).sum()), int(np.isinf(want).sum()), "equal_nan", bool(np.array_equal(got, want, equal_nan=True)))
        if bad.any():
            idx = np.argwhere(bad)[:3]
            for i in idx: print("   at", tuple(i), got[tuple(i)], want[tuple(i)], got[tuple(i)].tobytes().hex(), want[tuple(i)].tobytes().hex())
