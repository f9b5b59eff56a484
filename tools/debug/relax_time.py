import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pde_multigrid_amd as P
n = int(sys.argv[1]); dtype = np.float32 if sys.argv[2] == "f32" else np.float64
ctx = P.Context(0)
for k, v in [a.split("=") for a in os.environ.get("MGX_PARAMS", "").split(",") if a]:
    ctx.set_param(k, int(v))
mg = P.MultiGrid3D(ctx, [n] * 3, [0, 1, 0, 1, 0, 1], dtype, nlevels=1)
e0, e1 = ctx.event(), ctx.event()
mg.Relax(0, 2); ctx.sync(); ctx.record(e0)
for _ in range(10): mg.Relax(0, 2)
ctx.record(e1)
ms = ctx.elapsed_ms(e0, e1) / 40
print("%s %s per colour pass %.2f us  %.1f GB/s  frac %.3f  [%s]" % (n, sys.argv[2], ms * 1e3, 1.5 * (n - 2) ** 3 * np.dtype(dtype).itemsize / ms / 1e6, 1.5 * (n - 2) ** 3 * np.dtype(dtype).itemsize / ms / 1e6 / 8000, ctx.last_relax_kernel()))
