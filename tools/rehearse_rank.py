#!/usr/bin/python3
"""What would ONE of eight ranks do in the 1025^3 run, measured on the one GPU that is here?  The rank is rehearsed
(mgx_comm_init_rehearsal): its own 128-plane slab hierarchy, the real overlap schedule, every ghost plane, all-gather and
all-reduce at full size -- through RCCL, from this rank to itself.  The values received are wrong, so no result is checked;
what is measured is this rank's compute plus the launch / stream / RCCL call pattern, i.e. everything but the wire.
    python tools/rehearse_rank.py [n=1025] [nranks=8] [one_gpu_ms=0] [only_rank=-1]
only_rank >= 0: that rank alone with the library's default exchange modes (for a profiler; give one_gpu_ms > 0 to skip the reference run)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
nranks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
one_gpu_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
only_rank = int(sys.argv[4]) if len(sys.argv) > 4 else -1
R3 = [0, 1, 0, 1, 0, 1]
MIN_PLANES = int(os.environ.get("MGX_REHEARSE_MIN_PLANES", "32"))  # bench.py's --min-planes default
CA = os.environ.get("MGX_REHEARSE_CA")  # ca_min_planes of the hierarchy (unset: the library default)


def timed(ctx, mg, steps=10):
    for _ in range(3):
        mg.VCycle(0, 2, 2)
    ctx.sync()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            mg.VCycle(0, 2, 2)
        ctx.sync()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    return sorted(ts)[1]


if one_gpu_ms <= 0:
    ctx = P.Context(0)
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64)
    one_gpu_ms = timed(ctx, mg, 5)
    mg.close()
    ctx.close()
print("one GPU, the whole %d^3 hierarchy: %.3f ms per V(2,2) cycle -> perfect %d-way share %.3f ms" % (n, one_gpu_ms, nranks, one_gpu_ms / nranks))
rows = []
modes = ((None, "library default (inline_bytes = 96 MB)"), (0, "every level overlapped"), (1 << 40, "every level inline"),
         (-1, "library default, cycle replayed from a HIP graph"))
for inline_bytes, label in ((modes[0], modes[3]) if only_rank >= 0 else modes):
    for vr in ((only_rank,) if only_rank >= 0 else (0, nranks // 2, nranks - 1)):
        ctx = P.Context(0)
        for kv in filter(None, os.environ.get("MGX_PARAMS", "").split(",")):  # e.g. MGX_PARAMS=slab.edges_merged=0
            ctx.set_param(kv.split("=")[0], int(kv.split("=")[1]))
        ctx.comm_init_rehearsal(P.Context.unique_id(), vr, nranks)
        mg = P.DistMultiGrid3D(ctx, [n] * 3, R3, np.float64, min_planes=MIN_PLANES, inline_bytes=None if inline_bytes == -1 else inline_bytes,
                               use_graph=inline_bytes == -1, ca_min_planes=None if CA is None else int(CA))
        ex0 = mg.n_exchanges
        mg.VCycle(0, 2, 2)
        exch = mg.n_exchanges - ex0
        ms = timed(ctx, mg)
        nd = mg.numDist
        mg.close()
        ctx.close()
        rows.append((label, vr, ms))
        print("%-40s rank %d of %d (%d distributed levels, %d exchanges per cycle): %.3f ms per cycle -> speed-up bound %.2f x of %d" % (
            label, vr, nranks, nd, exch, ms, one_gpu_ms / ms, nranks), flush=True)
worst = {}
for label, vr, ms in rows:
    worst[label] = max(worst.get(label, 0.0), ms)
for label, ms in worst.items():
    print("slowest rehearsed rank, %s: %.3f ms -> at most %.2f x on %d GPUs before any wire time" % (label, ms, one_gpu_ms / ms, nranks))
