#!/usr/bin/python3
"""V(2,2) cycles launched eagerly against replayed from a HIP graph (use_graph), single GPU.
    python3 tools/graph_time.py"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P
for n, dt in ((513, np.float64), (257, np.float64), (513, np.float32)):
    for g in (0, 1, 0, 1):
        ctx = P.Context(0)
        mg = P.MultiGrid3D(ctx, [n]*3, [0,1,0,1,0,1], dt)
        mg.use_graph = bool(g)
        for _ in range(5): mg.VCycle(0, 2, 2)
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ts = []
        for b in range(5):
            ctx.record(e0)
            for _ in range(20): mg.VCycle(0, 2, 2)
            ctx.record(e1); ctx.sync()
            ts.append(ctx.elapsed_ms(e0, e1) / 20)
        print(n, np.dtype(dt).name, "graph" if g else "eager", "median %.4f ms min %.4f" % (sorted(ts)[2], min(ts)), flush=True)
        mg.close(); ctx.close()
