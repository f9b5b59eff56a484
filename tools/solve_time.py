#!/usr/bin/python3
"""End-to-end time of the drop-in entry mg3d_solve (host arrays in the reference layout in, solution out): hierarchy
construction, upload of grid + rhs (relayout to x-split on the device), cycles, download -- the PCIe-inclusive rate."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import pde_multigrid_amd as P
from pde_multigrid_amd._lib import lib, check
ctx = P.Context(0)
for n, dtype in ((257, np.float64), (513, np.float64), (513, np.float32)):
    x = np.linspace(0, 1, n)
    rhs = (-3 * np.pi ** 2 * np.sin(np.pi * x)[:, None, None] * np.sin(np.pi * x)[None, :, None] * np.sin(np.pi * x)[None, None, :]).astype(dtype)
    fn = getattr(lib, "mg3d_solve_" + ("f32" if dtype == np.float32 else "f64"))
    ct = C.c_float if dtype == np.float32 else C.c_double
    for ncyc in (1, 10):
        best = 1e9
        grid = np.empty((n, n, n), dtype)
        for rep in range(3):
            grid[...] = 0  # touched pages: a fresh np.zeros array would add ~25 ms of page faults per GB to the copies
            ctx.sync()
            t0 = time.perf_counter()
            # the C entry itself, in place on `grid` (the Python convenience wrapper P.solve3d copies its argument first)
            check(fn(ctx._h, grid.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p), (C.c_int * 3)(n, n, n),
                     (ct * 6)(0, 1, 0, 1, 0, 1), C.c_int(0), C.c_int(0), C.c_int(1), C.c_int(2), C.c_int(2), C.c_int(ncyc), C.c_int(0)))
            best = min(best, time.perf_counter() - t0)
        gb = 3 * grid.nbytes / 1e9
        print("mg3d_solve %d^3 %s, %2d V(2,2) cycles: %.1f ms end to end (%.2f GB over PCIe)" % (n, np.dtype(dtype).name, ncyc, best * 1e3, gb), flush=True)
