#!/usr/bin/python3
"""How many ghost exchanges of a slab V(2,2) cycle are EXPOSED (on the critical path) rather than overlapped: the
in-process test transport can delay every exchange by a fixed time; the slope of the cycle time over that delay is the
number of exposed exchange latencies per cycle (thread-ranks share one GPU here, so the absolute times are not those of
N GPUs -- the slope is what carries over: exposed exchanges x RCCL latency is the part of a cycle that does not scale).

    python3 tools/dist_exposure.py [n] [nranks] [min_planes]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_planes = int(sys.argv[3]) if len(sys.argv) > 3 else 16
R3 = [0, 1, 0, 1, 0, 1]
rows = []
for delay in [int(x) for x in os.environ.get("MGX_DELAYS", "0,250,500,1000").split(",")]:
    ctxs = [P.Context(0) for _ in range(nranks)]
    group = P.LocalGroup(nranks)
    group.set_test_hooks(delay, False)
    for r, c in enumerate(ctxs):
        group.attach(c, r)
    out = {}
    bar = threading.Barrier(nranks)

    def worker(r):
        ib = os.environ.get("MGX_INLINE_BYTES")
        mg = P.DistMultiGrid3D(ctxs[r], [n] * 3, R3, np.float64, min_planes=min_planes, inline_bytes=None if ib is None else int(ib))
        out["levels"] = (mg.numDist, mg.numGrids)
        for _ in range(2):
            mg.VCycle(0, 2, 2)
        ctxs[r].sync()
        bar.wait()
        t0 = time.perf_counter()
        for _ in range(5):
            mg.VCycle(0, 2, 2)
        ctxs[r].sync()
        out[r] = (time.perf_counter() - t0) / 5
        bar.wait()
        mg.close()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(nranks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for c in ctxs:
        c.close()
    group.close()
    ms = max(out[r] for r in range(nranks)) * 1e3
    rows.append((delay, ms))
    print("delay %4d us per exchange: %.3f ms per cycle  (levels distributed / all: %s)" % (delay, ms, out["levels"]), flush=True)
d0, t0 = rows[0]
for d, t in rows[1:]:
    print("  slope to delay %4d: %.1f exposed exchange latencies per cycle" % (d, (t - t0) * 1e3 / d))
