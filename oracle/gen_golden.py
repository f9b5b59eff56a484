#!/usr/bin/python3
"""Generate tests/golden/ from the UNMODIFIED compiled reference -- container-only.

Runs oracle/_ref/libmgref.so (= /root/reference's NOCUDA_TESI sources compiled in place by
`make -C oracle ref`, driven through oracle/ref_shim.cpp) on seeded inputs and stores
inputs + reference outputs as small .npz fixtures, plus known-answer hashes/values of the
larger BASELINE.json configurations in known_answers.json.

The fixtures are DATA (inputs and the reference's outputs); no reference source is stored.
Hash = 64-bit FNV-1a-style over the 32-bit patterns in memory order (oracle.fnv).

    python oracle/gen_golden.py          # rewrites tests/golden/
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as O  # noqa: E402  (only for fnv + shapes)
import refshim as R  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
A2 = [-1.0, -2.0, 0.0, -3.0]  # N2/LyapunovSolver.cpp:20-23
ALFA = 2                       # N2/LyapunovSolver.cpp:28


def u(rng, shape):
    return rng.uniform(-1.0, 1.0, shape).astype(np.float32)


def ops3d(n, seed):
    rng = np.random.default_rng(seed)
    n3 = [n, n, n]
    rg = [0.0, 1.0, 0.0, 2.0, 0.0, 3.0]  # anisotropic: hx != hy != hz
    v, f = u(rng, O.shape(n3)), u(rng, O.shape(n3))
    c = u(rng, O.shape(O.csize(n3)))
    d = dict(n=np.array(n3), range=np.array(rg, np.float32), v=v, f=f, c=c)
    d["relax1"] = R.relax3d(n3, rg, v, f, 1)
    d["relax3"] = R.relax3d(n3, rg, v, f, 3)
    d["residual"] = R.residual3d(n3, rg, v, f)
    d["restrict"] = R.restrict3d(n3, v)
    d["interpolate"] = R.interpolate3d(n3, v, c)
    d["correct"] = R.correct3d(n3, v, f)
    d["set_interior"] = R.set3d(n3, v, 2.5, False)
    d["set_all"] = R.set3d(n3, v, 2.5, True)
    d["vcycle22"] = R.cycle3d(n3, rg, mode=0, v1=2, v2=2, v=v, f=f)
    d["fmg122"] = R.cycle3d(n3, rg, mode=1, v0=1, v1=2, v2=2, v=v, f=f)
    return d


def ops2d(n, seed):
    rng = np.random.default_rng(seed)
    n2 = [n, n]
    rg = [0.0, 1.0, 0.0, 2.0]
    v, f = u(rng, O.shape(n2)), u(rng, O.shape(n2))
    c = u(rng, O.shape(O.csize(n2)))
    d = dict(n=np.array(n2), range=np.array(rg, np.float32), A=np.array(A2, np.float32),
             alfa=np.array(ALFA), v=v, f=f, c=c)
    d["relax1"] = R.relax2d(n2, rg, A2, ALFA, v, f, 1)
    d["relax3"] = R.relax2d(n2, rg, A2, ALFA, v, f, 3)
    d["residual"] = R.residual2d(n2, rg, A2, ALFA, v, f)
    d["restrict"] = R.restrict2d(n2, v)
    d["interpolate"] = R.interpolate2d(n2, v, c)
    d["correct"] = R.correct2d(n2, v, f)
    d["set_interior"] = R.set2d(n2, v, 2.5, False)
    d["set_all"] = R.set2d(n2, v, 2.5, True)
    d["vcycle22"] = R.cycle2d(n2, rg, A2, ALFA, mode=0, v1=2, v2=2, v=v, f=f)
    d["fmg122"] = R.cycle2d(n2, rg, A2, ALFA, mode=1, v0=1, v1=2, v2=2, v=v, f=f)
    return d


def ops1d(n, seed):
    rng = np.random.default_rng(seed)
    rg = [0.0, 1.0]
    v, f = u(rng, n), u(rng, n)
    c = u(rng, (n - 1) // 2 + 1)
    d = dict(n=np.array(n), range=np.array(rg, np.float32), v=v, f=f, c=c)
    d["relax1"] = R.relax1d(n, rg, v, f, 1)
    d["relax3"] = R.relax1d(n, rg, v, f, 3)
    d["residual"] = R.residual1d(n, rg, v, f)
    d["restrict"] = R.restrict1d(n, v)
    d["interpolate"] = R.interpolate1d(n, v, c)
    d["correct"] = R.correct1d(n, v, f)
    d["set_interior"] = R.set1d(n, v, 2.5, False)
    d["set_all"] = R.set1d(n, v, 2.5, True)
    d["vcycle22"] = R.cycle1d(n, rg, mode=0, v1=2, v2=2, v=v, f=f)
    d["fmg122"] = R.cycle1d(n, rg, mode=1, v0=1, v1=2, v2=2, v=v, f=f)
    return d


def rel_l2_3d(o, n):
    x = np.linspace(0.0, 1.0, n)
    s = np.sin(np.pi * x)
    uex = s[None, None, :] * s[None, :, None] * s[:, None, None]
    return float(np.linalg.norm(o.astype(np.float64) - uex) / np.linalg.norm(uex))


def known_answers():
    """Analytic-init runs of the unmodified reference."""
    ka = {"_hash": "FNV-1a-style over 32-bit words of finest h_v in memory order (oracle.fnv)",
          "_glibc_note": "analytic RHS uses libm sin/expf: valid for this image's glibc 2.35"}
    r3 = [0, 1, 0, 1, 0, 1]
    cases3 = [
        ("3d_n9_fmg122", dict(n=9, mode=1, v0=1, v1=2, v2=2)),
        ("3d_n17_fmg122", dict(n=17, mode=1, v0=1, v1=2, v2=2)),
        ("3d_n33_vcycle22", dict(n=33, mode=0, v1=2, v2=2)),
        ("3d_n65_vcycle22", dict(n=65, mode=0, v1=2, v2=2)),
        ("3d_n257_vcycle22_6lev", dict(n=257, mode=0, v1=2, v2=2, nlevels=6)),  # BASELINE config 3
        ("3d_n513_vcycle22_9lev", dict(n=513, mode=0, v1=2, v2=2)),  # bench.py's size, in the reference's own fp32 (1 min)
        ("3d_n17_fmg_2_3000_3000", dict(n=17, mode=1, v0=2, v1=3000, v2=3000)),  # thesis parameters
        ("3d_n129_relax10", dict(n=129, mode=0, v1=10, v2=0, nlevels=1)),
        ("3d_n257_relax4", dict(n=257, mode=0, v1=4, v2=0, nlevels=1)),
    ]
    for name, kw in cases3:
        n = kw.pop("n")
        o = R.cycle3d([n] * 3, r3, **kw)
        ka[name] = dict(n=n, **kw, hash=O.fnv(o), centre=float(o[n // 2, n // 2, n // 2]),
                        rel_l2_vs_analytic=rel_l2_3d(o, n))
        print(name, ka[name], flush=True)
    r2 = [0, 1, 0, 1]
    cases2 = [
        ("2d_n33_fmg122", dict(n=33, mode=1, v0=1, v1=2, v2=2)),
        ("2d_n129_fmg122", dict(n=129, mode=1, v0=1, v1=2, v2=2)),
        ("2d_n257_fmg_1_500_500", dict(n=257, mode=1, v0=1, v1=500, v2=500)),
        ("2d_n1025_vcycle22_7lev", dict(n=1025, mode=0, v1=2, v2=2, nlevels=7)),  # BASELINE config 2
        ("2d_n1025_relax20", dict(n=1025, mode=0, v1=20, v2=0, nlevels=1)),
    ]
    for name, kw in cases2:
        n = kw.pop("n")
        o = R.cycle2d([n] * 2, r2, A2, ALFA, **kw)
        ka[name] = dict(n=n, **kw, hash=O.fnv(o), centre=float(o[n // 2, n // 2]))
        print(name, ka[name], flush=True)
    cases1 = [
        ("1d_n4097_vcycle22_5lev", dict(n=4097, mode=0, v1=2, v2=2, nlevels=5)),  # BASELINE config 1
        ("1d_n4097_fmg122", dict(n=4097, mode=1, v0=1, v1=2, v2=2)),
        ("1d_n257_fmg_2_1000_1000", dict(n=257, mode=1, v0=2, v1=1000, v2=1000)),
        ("1d_n4097_fmg_2_1000_1000", dict(n=4097, mode=1, v0=2, v1=1000, v2=1000)),
    ]
    for name, kw in cases1:
        n = kw.pop("n")
        o = R.cycle1d(n, [0, 1], **kw)
        ka[name] = dict(n=n, **kw, hash=O.fnv(o), centre=float(o[n // 2]))
        print(name, ka[name], flush=True)
    return ka


def main():
    if not R.available():
        sys.exit("oracle/_ref/libmgref.so missing: run `make -C oracle ref` (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    for n, seed in ((5, 305), (9, 309), (17, 317)):
        np.savez_compressed(os.path.join(OUT, "ops3d_n%d.npz" % n), **ops3d(n, seed))
    for n, seed in ((9, 209), (17, 217), (33, 233)):
        np.savez_compressed(os.path.join(OUT, "ops2d_n%d.npz" % n), **ops2d(n, seed))
    for n, seed in ((17, 117), (65, 165)):
        np.savez_compressed(os.path.join(OUT, "ops1d_n%d.npz" % n), **ops1d(n, seed))
    # analytic-init inputs of small levels (so that tests on the GPU box do not depend on libm)
    v, f = R.init3d([17] * 3, [0, 1, 0, 1, 0, 1], 0)
    np.savez_compressed(os.path.join(OUT, "init3d_n17.npz"), v=v, f=f)
    v, f = R.init2d([33] * 2, [0, 1, 0, 1], A2, ALFA, 0)
    np.savez_compressed(os.path.join(OUT, "init2d_n33.npz"), v=v, f=f)
    v, f = R.init1d(65, [0, 1], 0)
    np.savez_compressed(os.path.join(OUT, "init1d_n65.npz"), v=v, f=f)
    with open(os.path.join(OUT, "known_answers.json"), "w") as fh:
        json.dump(known_answers(), fh, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
