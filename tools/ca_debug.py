"""scratch: locate mismatches of the CA slab schedule (thread-ranks) against the oracle"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as O
import pde_multigrid_amd as P
from test_gpu_dist import run_ranks

P.lib.mgx_test_set_lds_poison(1)
n3 = [33, 65, 257]
rg = [-1, 1, 0, 2, 0.5, 3]
rng = np.random.default_rng(102)
v0 = rng.uniform(-1, 1, O.shape(n3))
f0 = rng.uniform(-1, 1, O.shape(n3))
for nranks, inl, v1, v2, cyc, minp in [(4, 0, 0, 2, 1, 16), (4, None, 0, 2, 1, 16), (4, 0, 0, 2, 1, 64), (4, 0, 0, 2, 1, 32), (4, 0, 0, 2, 2, 64), (2, 0, 0, 2, 1, 64), (4, 0, 0, 1, 1, 64)]:
    got, info = run_ranks(nranks, n3, rg, np.float64, v1, v2, cyc, minp, v0=v0, f0=f0, inline_bytes=inl)
    want = O.cycle3d(n3, rg, mode=0, v1=v1, v2=v2, reps=cyc, v=v0, f=f0, dtype=np.float64)
    bad = [z for z in range(n3[2]) if not np.array_equal(got[z].view(np.uint64), want[z].view(np.uint64))]
    print("ranks %d inline %s V(%d,%d) x%d min_planes %d levels %s: %d bad planes %s" % (nranks, inl, v1, v2, cyc, minp, info[0], len(bad), bad[:40]), flush=True)
    for z in bad[:2]:
        d = got[z].view(np.uint64) != want[z].view(np.uint64)
        ys, xs = np.nonzero(d)
        print("   plane %d: %d bad points, y %d..%d x %d..%d, parity of x+y+z: %s, nan %d" % (z, d.sum(), ys.min(), ys.max(), xs.min(), xs.max(),
              sorted(set(((xs + ys + z) & 1).tolist())), int(np.isnan(got[z]).sum())), flush=True)
