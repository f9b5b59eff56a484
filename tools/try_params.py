#!/usr/bin/python3
"""A/B a set of context parameters on the smoother (interleaved rounds, one process).
    python tools/try_params.py "relax3d.nt=0" "relax3d.nt=1" "relax3d.nt=2,relax3d.zchunk=8" ... [--n 513]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = 513
for a in sys.argv[1:]:
    if a.startswith("--n="):
        n = int(a[4:])
ctx = P.Context(0)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float64, nlevels=2)
e0, e1 = ctx.event(), ctx.event()
base = {"relax3d.ty": 4, "relax3d.rows": 4, "relax3d.zchunk": 0, "relax3d.xcd": 1, "relax3d.ablate": 0, "relax3d.lds": -1,
        "residual_restrict3d.stream": 3, "residual_restrict3d.pzchunk": 0, "residual_restrict3d.cr": 0,
        "residual_restrict3d.tyw": 4}
res = {a: [] for a in args}
rr = {a: [] for a in args}
for rnd in range(4):
    for a in args:
        for k, v in base.items():
            ctx.set_param(k, v)
        for kv in a.split(","):
            k, v = kv.split("=")
            ctx.set_param(k, int(v))
        ctx.sync(); ctx.record(e0)
        for _ in range(5):
            mg.Relax(0, 2)
        ctx.record(e1)
        t = ctx.elapsed_ms(e0, e1) / 10
        mg.numGrids = 2
        ctx.sync(); ctx.record(e0)
        for _ in range(3):
            mg._call("VCycle", 0, 0, 0)   # residual+restrict, zero, (coarse: nothing), interpolate+correct
        ctx.record(e1)
        t2 = ctx.elapsed_ms(e0, e1) / 3
        if rnd:
            res[a].append(t); rr[a].append(t2)
for a in args:
    print("%-60s relax %.4f ms/sweep (%.1f GLUPS)   rr+interp %.4f ms" % (a, np.median(res[a]), (n - 2) ** 3 / np.median(res[a]) / 1e6, np.median(rr[a])))
