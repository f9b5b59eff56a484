#!/usr/bin/python3
"""Where a step of the one-launch sweep kernel spends its cycles (diagnostic library: make -C pde_multigrid_amd/csrc diag):
per-wave cycle stamps of the phases, and ablations (WRONG results) timed with HIP events.
    python tools/sweep_stamps.py [n] [lead]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MGX_LIB_PATH", os.path.join(ROOT, "pde_multigrid_amd", "lib", "libmgx_diag.so"))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
lead = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = P.Context(0)
ctx.set_param("relax3d.fused", 1)
ctx.set_param("relax3d.fused_lead", lead)
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], np.float64, nlevels=2)
e0, e1 = ctx.event(), ctx.event()


def timed(fn, reps=10):
    ts = []
    for i in range(reps + 2):
        ctx.sync()
        ctx.record(e0)
        fn()
        ctx.record(e1)
        ts.append(ctx.elapsed_ms(e0, e1))
    return sorted(ts[2:])[len(ts[2:]) // 2]


NAMES = ["issue", "flagwait", "red", "black", "barrier", "vmcnt0", "-", "-"]
ABL = [(0, "full"), (1, "no flag waits"), (2, "no sc1"), (3, "no waits, no sc1"), (4, "no black arithmetic"), (8, "no red arithmetic"),
       (12, "no arithmetic"), (16, "no stores"), (31, "nothing but loads + LDS")]
for bits, name in ABL:
    ctx.set_param("relax3d.fused_dbg", 1 + 2 * bits)
    ms = timed(lambda: mg.Relax(0, 2))
    nwg, nw = 256, 8
    buf = (C.c_longlong * (nwg * nw * 8))()
    P.check(P.lib.mgx_sweep_debug_read(ctx._h, buf, C.c_size_t(nwg * nw * 8)))
    a = np.frombuffer(buf, dtype=np.int64).reshape(nwg, nw, 8).astype(np.float64)
    tot = a.sum(axis=2)
    print("%-26s Relax(2) %.4f ms = %.4f per sweep; cycles per wave %.0f (min %.0f max %.0f)" % (name, ms, ms / 2, tot.mean(), tot.min(), tot.max()))
    print("    " + "  ".join("%s %.1f%%" % (NAMES[k], 100 * a[:, :, k].sum() / tot.sum()) for k in range(6)))
    if bits == 0:
        for wv in range(nw):
            print("      wave %d: " % wv + "  ".join("%s %.1f%%" % (NAMES[k], 100 * a[:, wv, k].sum() / tot[:, wv].sum()) for k in range(6)))
ctx.set_param("relax3d.fused_dbg", 0)
mg.close()
