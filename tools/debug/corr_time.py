import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P
ctx = P.Context(0)
mg = P.MultiGrid3D(ctx, [513] * 3, [0, 1, 0, 1, 0, 1], np.float64)
for fuse in (1, 0, 1, 0):
    ctx.set_param("relax3d.corr_fuse", fuse)
    for _ in range(5):
        mg.VCycle(0, 2, 2)
    ctx.sync()
mg.close(); ctx.close()
