#!/usr/bin/python3
"""Tuning sweep of the 3D smoother on one MI355X: interleaved rounds in ONE process
(cdna_hip_programming.md rule 24), HIP-event timing on the compute stream.

    python tools/sweep_relax.py [--n 513] [--dtype f64] [--sweeps 10] [--rounds 5]
Prints one line per configuration: median / min ms per red+black sweep, MLUPS, algorithmic GB/s
(3 reals per update) and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_multigrid_amd as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=513)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--sweeps", type=int, default=2, help="sweeps per Relax call (V(2,2) calls Relax with 2)")
    ap.add_argument("--calls", type=int, default=5, help="Relax calls per timing")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--configs", default="")
    args = ap.parse_args()
    dtype = np.float64 if args.dtype == "f64" else np.float32
    w = np.dtype(dtype).itemsize
    ctx = P.Context(0)
    n = args.n
    cfgs = [("natural", 4, 4, 0, 0, 0, 0)]
    if args.configs:
        for c in args.configs.split(","):
            ty, rows, zc, xcd, wp, sh = c.split(":")
            cfgs.append(("xsplit", int(ty), int(rows), int(zc), int(xcd), int(wp), int(sh)))
    else:
        for sh in (1, 0):
            for ty, rows in ((4, 2), (4, 4), (2, 4), (4, 1), (8, 1), (8, 2), (2, 8), (4, 8)):
                for zc in (4, 8, 16):
                    cfgs.append(("xsplit", ty, rows, zc, 1, 0, sh))
    mgs = {lay: P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype, nlevels=1, layout=lay) for lay in ("natural", "xsplit")}
    e0, e1 = ctx.event(), ctx.event()
    times = {c: [] for c in cfgs}
    for r in range(args.rounds + 1):
        for c in cfgs:
            lay, ty, rows, zc, xcd, wp, sh = c
            ctx.set_param("relax3d.ty", ty)
            ctx.set_param("relax3d.rows", rows)
            ctx.set_param("relax3d.wave_planes", wp)
            ctx.set_param("relax3d.zchunk", zc)
            ctx.set_param("relax3d.xcd", xcd)
            mg = mgs[lay]
            ctx.sync()
            ctx.record(e0)
            for _ in range(args.calls):
                mg.Relax(0, args.sweeps)
            ctx.record(e1)
            ms = ctx.elapsed_ms(e0, e1) / (args.sweeps * args.calls)
            if r > 0:
                times[c].append(ms)
    lups = (n - 2) ** 3
    out = []
    for c in cfgs:
        t = np.array(times[c])
        med, mn = float(np.median(t)), float(t.min())
        gbs = 3 * w * lups / (med * 1e-3) / 1e9
        out.append(dict(layout=c[0], ty=c[1], rows=c[2], zchunk=c[3], xcd=c[4], wave_planes=c[5], shfl=c[6], ms_median=round(med, 4), ms_min=round(mn, 4),
                         mlups=round(lups / (med * 1e-3) / 1e6, 1), alg_GBps=round(gbs, 1), frac_hbm=round(gbs / 8000.0, 4)))
    out.sort(key=lambda r: r["ms_median"])
    for r in out:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
