#!/usr/bin/python3
"""Run-to-run and kernel-to-kernel reproducibility at the bench size: 30 V(2,2) cycles at 513^3 fp64, twice with the
default kernels and once with the previous generation (relax3d.lds = 0, residual_restrict3d.stream = 1); the three
results must be bit-identical (a race in an in-place update would show up here long before it does in a short test)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import oracle as O  # noqa: E402  (hash only)
import pde_multigrid_amd as P  # noqa: E402

ctx = P.Context(0)
hashes = []
for name, params in (("default", {}), ("default again", {}), ("previous kernels", {"relax3d.lds": 0, "residual_restrict3d.stream": 1})):
    for k, v in params.items():
        ctx.set_param(k, v)
    mg = P.MultiGrid3D(ctx, [513] * 3, [0, 1, 0, 1, 0, 1], np.float64, residual_mode=P.CORRECT)
    for _ in range(30):
        mg.VCycle(0, 2, 2)
    v = mg.download_v(0)
    hashes.append(O.fnv(v))
    print(name, hashes[-1], "rel. L2 error vs analytic %.3e" % mg.DiffStats(0)[2])
    mg.close()
assert len(set(hashes)) == 1, hashes
print("bit-identical")
