#!/usr/bin/python3
"""TIMING ONLY (diagnostic library, wrong results): what is left of the unrolled plain colour pass when parts of a step are switched off --
"relax3d.ablate" = 100 + bits: 1 the update's arithmetic (seven additions instead), 2 the barrier, 4 the stores, 8 the column / f loads.
    python3 tools/pipe_ablate.py [n=513] [f64|f32]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MGX_LIB_PATH", os.path.join(ROOT, "pde_multigrid_amd", "lib", "libmgx_diag.so"))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
dtype = np.float32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else np.float64
ctx = P.Context(0)
if dtype == np.float32:
    ctx.set_param("relax3d.v2", 0)
    ctx.set_param("relax3d.unroll", 15)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype)
e0, e1 = ctx.event(), ctx.event()


def timed(reps=12):
    ts = []
    for i in range(reps + 3):
        ctx.sync()
        ctx.record(e0)
        mg.Relax(0, 2)
        ctx.record(e1)
        ts.append(ctx.elapsed_ms(e0, e1))
    ts = sorted(ts[3:])
    return ts[len(ts) // 2]


NAMES = {0: "the pass as it is", 1: "no arithmetic", 2: "no barrier", 3: "no arithmetic, no barrier", 4: "no stores", 8: "no column / f loads",
         12: "no loads, no stores (arithmetic, LDS, barrier only)", 13: "no loads, no stores, no arithmetic (LDS + barrier + loop)", 7: "loads only (no arithmetic, barrier, stores)"}
for bits in (0, 1, 2, 3, 4, 8, 12, 13, 7, 0):
    ctx.set_param("relax3d.ablate", 100 + bits if bits else 0)
    print("n=%d %s: %-60s %.1f us per pass" % (n, np.dtype(dtype).name, NAMES[bits], timed() * 250), flush=True)
