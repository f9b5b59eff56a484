#!/usr/bin/python3
"""Ablation of the x-split smoother kernel (diagnostic builds, WRONG results): which resource bounds it?
    python tools/ablate_relax.py [--n 513]
mask bits: 1 = no f load, 2 = no store, 4 = no side/edge loads (only the streaming U load), 8 = no division,
16 = plain instead of non-temporal stores (the only variant with correct results)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the ablation variants live in the diagnostic twin only: make -C pde_multigrid_amd/csrc diag
os.environ.setdefault("MGX_LIB_PATH", os.path.join(ROOT, "pde_multigrid_amd", "lib", "libmgx_diag.so"))
import pde_multigrid_amd as P  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=513)
args = ap.parse_args()
n = args.n
ctx = P.Context(0)
ctx.set_param("relax3d.ty", 4)
ctx.set_param("relax3d.rows", 4)
ctx.set_param("relax3d.wave_planes", 0)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float64, nlevels=1)
e0, e1 = ctx.event(), ctx.event()
names = {0: "full kernel", 1: "no f load", 2: "no store", 3: "no f, no store", 4: "no side/edge loads", 5: "no f, no side/edge",
         7: "only the U stream (no f, store, side/edge)", 8: "no division", 12: "no side/edge, no division", 15: "U stream only, no division", 16: "plain (temporal) stores"}
for rnd in range(3):
    for mask in (0, 16, 1, 2, 3, 4, 5, 7, 8, 12, 15):
        ctx.set_param("relax3d.ablate", mask)
        ctx.sync()
        ctx.record(e0)
        for _ in range(5):
            mg.Relax(0, 2)
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 10
        if rnd == 2:
            print("mask %2d  %-46s %.4f ms/sweep  %6.1f GLUPS" % (mask, names[mask], ms, (n - 2) ** 3 / ms / 1e6))
