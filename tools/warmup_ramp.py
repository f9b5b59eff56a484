#!/usr/bin/python3
"""How long does the GPU take to reach its steady state?  513^3 fp64 V(2,2) cycles from a cold start (fresh process, hierarchy just
built), timed in groups of 10 with HIP events on the compute stream.  bench.py's default warm-up (150 cycles) comes from this.
    python3 tools/warmup_ramp.py [n=513] [cycles=400]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 400
ctx = P.Context(0)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float64)
G = 10
evs = [ctx.event() for _ in range(cycles // G + 1)]
ctx.sync()
ctx.record(evs[0])
for g in range(cycles // G):
    for _ in range(G):
        mg.VCycle(0, 2, 2)
    ctx.record(evs[g + 1])
ctx.sync()
ms = [ctx.elapsed_ms(evs[g], evs[g + 1]) / G for g in range(cycles // G)]
last = float(np.median(ms[-10:]))
print("V(2,2) %d^3 fp64 from a cold start, ms per cycle in groups of %d (steady state = median of the last 100 cycles: %.4f ms):" % (n, G, last))
for g, t in enumerate(ms):
    print("  cycles %4d-%4d: %.4f ms  (%+.1f %%)" % (g * G, g * G + G - 1, t, (t / last - 1) * 100))
mg.close()
ctx.close()
