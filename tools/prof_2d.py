#!/usr/bin/python3
"""workload for rocprofv3: 50 V(2,2) cycles of the 2D Lyapunov 1025^2 7-level hierarchy (fp64, cache-resident kernels)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

ctx = P.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
mg = P.MultiGrid2D(ctx, [n] * 2, [0, 1, 0, 1], [-1, -2, 0, -3], 2, np.float64, nlevels=7 if n == 1025 else 0)
for _ in range(50):
    mg.VCycle(0, 2, 2)
ctx.sync()
mg.close()
ctx.close()
