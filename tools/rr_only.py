#!/usr/bin/python3
"""Run only residual+restrict + interpolate+correct of the finest level (for rocprofv3 --pmc passes).
    python3 tools/rr_only.py [--n=513] [name=value,...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n = 513
ctx = P.Context(0)
for a in sys.argv[1:]:
    if a.startswith("--n="):
        n = int(a[4:])
    else:
        for kv in a.split(","):
            k, v = kv.split("=")
            ctx.set_param(k, int(v))
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float64, nlevels=2)
mg.Relax(0, 1)
for _ in range(6):
    mg._call("VCycle", 0, 0, 0)
ctx.sync()
mg.close()
ctx.close()
