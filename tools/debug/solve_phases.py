import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pde_multigrid_amd as P
ctx = P.Context(0)
n = 513
v = np.zeros((n, n, n)); f = np.ones((n, n, n))
def T(name, fn):
    ctx.sync(); t0 = time.perf_counter(); r = fn(); ctx.sync(); print("%-14s %.1f ms" % (name, (time.perf_counter() - t0) * 1e3), flush=True); return r
for rep in range(2):
    mg = T("create", lambda: P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float64))
    T("upload_v", lambda: mg.upload_v(0, v))
    T("upload_f", lambda: mg.upload_f(0, f))
    T("VCycle", lambda: mg.VCycle(0, 2, 2))
    out = T("download_v", lambda: mg.download_v(0))
    T("close", lambda: mg.close())
