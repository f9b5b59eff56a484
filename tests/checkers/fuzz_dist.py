#!/usr/bin/python3
"""One-off fuzz of the z-slab driver (thread-ranks on one GPU over the asynchronous test transport, delay hook on) against
the single-GPU hierarchy: random shapes, rank counts, agglomeration thresholds, sweep counts, modes, V-cycles and FMG.

    python3 tests/checkers/fuzz_dist.py [cases] [seed] [zmin]
zmin: the z size of every case is at least that (129 / 257: slabs thick enough for the communication-avoiding schedule)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O  # noqa: E402
import pde_multigrid_amd as P  # noqa: E402
from test_gpu_dist import run_ranks  # noqa: E402

# MGX_PARAMS=name=value,... : library parameters of the ranks' contexts (e.g. rr3d.black=2: the fused way down on slabs)
PARAMS = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in filter(None, os.environ.get("MGX_PARAMS", "").split(","))}

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
zmin = int(sys.argv[3]) if len(sys.argv) > 3 else 0
bad = 0
for c in range(cases):
    dtype = np.float64 if rng.random() < 0.6 else np.float32
    nr = int(rng.choice([1, 2, 4, 8]))
    mp = int(rng.choice([2, 4, 8, 16]))
    while True:  # the finest level must be distributable: an even number >= min_planes of planes per rank
        n = [int(rng.choice([9, 17, 33, 65, 129, 257])) for _ in range(3)]
        if n[2] < zmin:
            n[2] = int(rng.choice([s for s in (129, 257) if s >= zmin]))
        if np.prod(n) <= 6e6 and (n[2] - 1) % nr == 0 and (n[2] - 1) // nr >= max(2, mp) and ((n[2] - 1) // nr) % 2 == 0:
            break
    box = [0.0, float(rng.choice([1.0, 2.0, 1.5])), 0.0, float(rng.choice([1.0, 3.0])), 0.0, float(rng.choice([1.0, 0.5]))]
    v1, v2 = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    mode = int(rng.integers(0, 2))
    fmg = int(rng.random() < 0.3) * 2
    reps = 0 if fmg else int(rng.integers(1, 3))
    v = rng.uniform(-1, 1, O.shape(n)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n)).astype(dtype)
    ctx = P.Context(0)
    mg = P.MultiGrid3D(ctx, n, box, dtype, residual_mode=mode)  # the reference run: library defaults
    mg.upload_f(0, f)
    mg.upload_v(0, v)
    if fmg:
        mg.FullMultiGridVCycle(0, fmg, v1, v2)
    for _ in range(reps):
        mg.VCycle(0, v1, v2)
    want = mg.download_v(0)
    mg.close()
    ctx.close()
    try:
        ib = [0, None, 200_000, 2_000_000][int(rng.integers(0, 4))]  # overlapped / library default / mixed thresholds
        got, info = run_ranks(nr, n, box, dtype, v1, v2, reps, mp, mode=mode, v0=v, f0=f, fmg=fmg, delay_us=200, join_timeout=120,
                              inline_bytes=ib, params=PARAMS)
        ok = got.tobytes() == want.tobytes()
        what = "levels dist/all %s inline_bytes=%s" % (info[0], ib)
    except AssertionError as e:
        ok, what = False, str(e)[:200]
    bad += not ok
    print("%s n=%s ranks=%d min_planes=%d %s V(%d,%d)x%d fmg=%d mode=%d %s" % ("ok  " if ok else "FAIL", n, nr, mp, np.dtype(dtype).name, v1, v2, reps, fmg, mode, what), flush=True)
print("%d cases, %d failures" % (cases, bad))
sys.exit(1 if bad else 0)
