#!/usr/bin/python3
"""bench.py -- MLUPS of the 3D Poisson V(2,2) cycle on MI355X, with the smoother's HBM roofline.

    python bench.py --gpus N --steps K --warmup W

N > 1 may be started either by a launcher (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`:
only the RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* it puts into the environment are used) or plainly: without WORLD_SIZE
in the environment this process only counts the devices (it never initialises the GPU, never imports libmgx and never
exec's), starts the N ranks itself as fresh child processes, watches them and exits with their code.  Either way the
ranks never import PyTorch: the control plane (the 128-byte ncclUniqueId from rank 0, barriers, the maximum over ranks)
is pde_multigrid_amd/launch.py, one TCP connection per rank on 127.0.0.1; the data plane is RCCL inside libmgx.

Workload
  N = 1 : BASELINE.json configs[3]: 3D Poisson, 513 points per axis ("512^3"), fp64, native 9-level
          hierarchy, analytic RHS of the reference (Grid3D::InitF), v = 0, reference (REF_COMPAT) semantics.
  N > 1 : BASELINE.json configs[4]: the same problem at 1025 points per axis ("1024^3"), ONE hierarchy
          decomposed into z-slabs over the N GPUs (ghost planes over RCCL/xGMI, coarse levels replicated after an
          all-gather) -- strong scaling: the total work does not depend on N.  `--size` overrides the size.
One step = one VCycle(0, 2, 2) through the C host layer (include/mg_multigrid.h), inputs resident in HBM.
    MLUPS = (v1+v2) * sum_levels (n_l - 2)^3 * steps / seconds          
    The K steps are timed BATCHES times (default 5; each batch = exactly K steps between a barrier + sync on both sides,
    maximum over ranks); `value` / `ms_per_step` are the MEDIAN batch, `batches` lists them all.  W untimed cycles come first
    (default 150 = 0.5 s at 513^3: the GPU needs ~100 cycles to reach its steady state, the first ones run 2-3 % slower).
secondary (N = 1, default size only): BASELINE configs[1] (2D Lyapunov 1025^2, 7 levels, fp64), configs[2] (3D 257^3, 6
    levels, fp64) and the headline workload in the reference's own precision (513^3 fp32), each timed the same way and
    checked against its committed known answer; and the reference's two PUBLISHED GPU workloads with their own parameters
    (3D FMG(2,3000,3000) at 129^3, 2D Lyapunov FMG(2,500,500) at 4097^2; whole programs, fp32), checked
    the same way.
roofline: the dominant kernel is the red-black Gauss-Seidel smoother on the finest level; its algorithmic
    traffic is 3 reals per lattice update per red+black sweep = 24 B/LUP in fp64 (12 B per LUP of one colour
    launch).  `achieved` = algorithmic bytes per launch / average launch duration, measured here with HIP events
    on the stream the kernel runs on, over a smoother-only timed region (rank 0's slab when N > 1, without the
    ghost exchange).  `kernel` is the name the library reports for the launch; `traffic` (HBM bytes per launch from
    separate rocprofv3 --pmc passes, profiles/pmc_traffic.json) is attached only when it was taken from that kernel.
result check: after the timed regions one more cycle is run from v = 0 and the finest v is compared with the
    committed known answer of the oracle (tests/golden/known_answers_f64.json: a position-weighted 64-bit checksum
    that numpy can evaluate); on a mismatch NO metric line is printed (exit 3).
cpu_baseline: the NOCUDA_TESI CPU path on this box's host cores, 1 thread, bounded sample, rank 0 at N = 1 only:
    the compiled reference itself (oracle/_ref, kind "reference") when it travelled with the repo, else the
    oracle's restatement (kind "port"); both optimisation levels (the reference's own CompileAndLink uses none),
    V(2,2) on 257^3 (6 levels) and the smoother on 513^3, plus the restatement/reference time ratio.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_BPS = 8.0e12  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8 TB/s; about 6.3 TB/s achievable)
R3 = [0, 1, 0, 1, 0, 1]


def level_sizes(n, nlevels):
    out = []
    for _ in range(nlevels):
        out.append(n)
        n = (n - 1) // 2 + 1
    return out


def checksum(a, first_word=0):
    """(sum w_i, sum w_i * (2 i + 1)) mod 2^64 over the words of `a` (64-bit words for fp64, 32-bit for fp32) in
    memory order, i counted from `first_word` (so that slabs of one array can be summed separately and added) --
    the same function as oracle/gen_known_f64.py, restated here because the product side of the bench does not
    import oracle/"""
    import numpy as np
    w = np.ascontiguousarray(a).reshape(-1)
    w = w.view(np.uint64) if w.dtype.itemsize == 8 else w.view(np.uint32)
    s1, s2, step = np.uint64(0), np.uint64(0), 1 << 24
    with np.errstate(over="ignore"):
        for i in range(0, w.size, step):
            c = w[i:i + step].astype(np.uint64)
            k = np.arange(first_word + i, first_word + i + c.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            s1 = s1 + c.sum(dtype=np.uint64)
            s2 = s2 + (c * k).sum(dtype=np.uint64)
    return int(s1), int(s2)


def known_answer(n, nlev, dtype):
    path = os.path.join(ROOT, "tests", "golden", "known_answers_f64.json")
    if not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh).get("3d_n%d_vcycle22_%dlev_%s" % (n, nlev, dtype))


def load_launch():
    """pde_multigrid_amd/launch.py WITHOUT importing the package (its __init__ loads libmgx, which the parent of a plain
    `bench.py --gpus N` must not do)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mgx_launch", os.path.join(ROOT, "pde_multigrid_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def spawn_ranks(args):
    """parent of a plain `bench.py --gpus N`: start the ranks as fresh children BEFORE anything touches the GPU here"""
    return load_launch().spawn(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=float(os.environ.get("MGX_BENCH_TIMEOUT", "1500")))


def cpu_baseline(args, dtype):
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O  # CPU baseline leg only: the checker timed as a reported baseline
    import refshim as RS
    cn, clev = 257, 6
    t_start = time.perf_counter()
    c_lups = (args.v1 + args.v2) * sum((s - 2) ** 3 for s in level_sizes(cn, clev))
    s_lups = (513 - 2) ** 3
    legs = {}

    def leg(name, secs, lups):
        legs[name] = {"seconds": round(secs, 3), "mlups": round(lups / secs / 1e6, 2)}
        return legs[name]["mlups"]

    have_ref = RS.available("O2") and RS.available("O0")
    # the restatement (same loop nest as the reference, float = the reference's type), both optimisation levels
    # (legs report seconds per repetition; 3 repetitions of the headline leg, 1 of the others: about 15 s on the GPU box's host)
    leg("port_f32_O2_vcycle257", O.time_vcycle3d(cn, clev, args.v1, args.v2, 3, np.float32, "O2") / 3, c_lups)
    leg("port_f32_O0_vcycle257", O.time_vcycle3d(cn, clev, args.v1, args.v2, 1, np.float32, "O0"), c_lups)
    leg("port_f32_O2_relax513", O.time_relax3d(513, 1, np.float32, "O2"), s_lups)
    leg("port_%s_O2_vcycle257" % args.dtype, O.time_vcycle3d(cn, clev, args.v1, args.v2, 1, dtype, "O2"), c_lups)
    if have_ref:
        value = leg("reference_f32_O2_vcycle257", RS.time_vcycle3d(cn, clev, args.v1, args.v2, 3, "O2") / 3, c_lups)
        leg("reference_f32_O0_vcycle257", RS.time_vcycle3d(cn, clev, args.v1, args.v2, 1, "O0"), c_lups)
        leg("reference_f32_O2_relax513", RS.time_relax3d(513, 1, "O2"), s_lups)
        leg("reference_f32_O0_relax513", RS.time_relax3d(513, 1, "O0"), s_lups)
        kind = "reference"
        what = "the compiled NOCUDA_TESI reference (oracle/_ref, fp32, g++ -O2; the -O0 legs are what its own CompileAndLink builds)"
        ratio = round(legs["port_f32_O2_vcycle257"]["seconds"] / legs["reference_f32_O2_vcycle257"]["seconds"], 3)
    else:
        leg("port_f32_O0_relax513", O.time_relax3d(513, 1, np.float32, "O0"), s_lups)
        value = legs["port_f32_O2_vcycle257"]["mlups"]
        kind = "port"
        what = "the oracle's CPU restatement of the reference (fp32, g++ -O2, reference loop nest)"
        ratio = None
    return {
        "value": value, "unit": "MLUPS", "cores": 1, "kind": kind,
        "sample": "3 V(%d,%d) cycles, 3D Poisson %d^3 fp32, %d levels, %s, 1 thread of %d host cores; all legs together %.1f s of CPU"
                  % (args.v1, args.v2, cn, clev, what, os.cpu_count() or 0, time.perf_counter() - t_start),
        "legs": legs,
        "port_over_reference_time": ratio,
    }


def run_secondary(P, ctx, np):
    """BASELINE configs[1], configs[2] and the headline workload in fp32, each: median of 3 batches of cycles from v = 0
    kept running (as the headline), then ONE cycle from v = 0 checked against the committed known answer"""
    path = os.path.join(ROOT, "tests", "golden", "known_answers_f64.json")
    with open(path) as fh:
        known = json.load(fh)
    out = {}

    def run(name, key, make, dim, nsz, nlev, steps, smoother_bytes=0):
        mg = make()
        ka = known.get(key)
        mg.VCycle(0, 2, 2)
        roof = None
        if smoother_bytes:  # the smoother of this hierarchy's finest level alone, as the headline's `roofline` is taken (HIP events)
            e0, e1 = ctx.event(), ctx.event()
            mg.Relax(0, 2)
            ctx.sync()
            ctx.record(e0)
            for _ in range(10):
                mg.Relax(0, 2)
            ctx.record(e1)
            launch_s = ctx.elapsed_ms(e0, e1) * 1e-3 / 40  # 10 calls x 2 sweeps x 2 colours
            per_launch = 3 * smoother_bytes * (nsz - 2) ** 3 / 2.0
            roof = {"bound": "hbm", "kernel": ctx.last_relax_kernel(), "avg_launch_us": round(launch_s * 1e6, 2),
                    "algorithmic_bytes_per_launch": per_launch, "achieved": round(per_launch / launch_s / 1e9, 1), "peak": HBM_PEAK_BPS / 1e9,
                    "unit": "GB/s", "frac": round(per_launch / launch_s / HBM_PEAK_BPS, 4), "traffic": None}
        ts = []
        for _ in range(3):
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                mg.VCycle(0, 2, 2)
            ctx.sync()
            ts.append((time.perf_counter() - t0) / steps)
        mg.close()
        mg = make()  # a fresh hierarchy: the reference's initial state (v = 0; 2D: its boundary values, N2/Grid2D.cpp:50-68)
        mg.VCycle(0, 2, 2)
        got = mg.download_v(0)
        s1, s2 = checksum(got)
        mg.close()
        lups = 4 * sum((s - 2) ** dim for s in level_sizes(nsz, nlev))
        t = sorted(ts)[1]
        status = "no known answer" if ka is None else ("ok" if ("%016x" % s1, "%016x" % s2) == (ka["sum64"], ka["wsum64"]) else "MISMATCH")
        out[name] = {"ms_per_cycle": round(t * 1e3, 4), "min_ms": round(min(ts) * 1e3, 4), "mlups": round(lups / t / 1e6, 1),
                     "steps": steps, "result_check": status, "known_answer": key}
        if roof:
            out[name]["roofline"] = roof

    run("configs[1]: 2D Lyapunov 1025^2, 7 levels, f64, V(2,2)", "2d_n1025_vcycle22_7lev_f64",
        lambda: P.MultiGrid2D(ctx, [1025] * 2, [0, 1, 0, 1], [-1, -2, 0, -3], 2, np.float64, nlevels=7), 2, 1025, 7, 200)
    run("configs[2]: 3D Poisson 257^3, 6 levels, f64, V(2,2)", "3d_n257_vcycle22_6lev_f64",
        lambda: P.MultiGrid3D(ctx, [257] * 3, R3, np.float64, nlevels=6), 3, 257, 6, 50)
    run("3D Poisson 513^3, 9 levels, f32 (the reference's precision), V(2,2)", "3d_n513_vcycle22_9lev_f32",
        lambda: P.MultiGrid3D(ctx, [513] * 3, R3, np.float32), 3, 513, 9, 20, smoother_bytes=4)
    # the reference's PUBLISHED workload (thesis Fig. 4.4 = BASELINE.md section 1: whole-program wall time, fp32): construction + RHS +
    # FMG(2, 3000, 3000) + download, n = 129; the thesis' GPU (GeForce GTX 550 Ti) took 39.1 s, its CPU run stopped at n = 65 (213.4 s)
    key = "3d_n129_fmg_2_3000_3000_f32"
    ka = known.get(key)
    ctx.sync()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        mg = P.MultiGrid3D(ctx, [129] * 3, R3, np.float32)
        mg.FullMultiGridVCycle(0, 2, 3000, 3000)
        got = mg.download_v(0)
        mg.close()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    s1, s2 = checksum(got)
    updates = 6000 * 2 * sum((j + 1) * (sz - 2) ** 3 for j, sz in enumerate(level_sizes(129, 7)))  # BASELINE.md section 1's count
    status = "no known answer" if ka is None else ("ok" if ("%016x" % s1, "%016x" % s2) == (ka["sum64"], ka["wsum64"]) else "MISMATCH")
    out["published workload: 3D Poisson FMG(2,3000,3000) 129^3, 7 levels, f32, whole program"] = {
        "seconds": round(min(ts), 4), "runs_s": [round(t, 4) for t in ts], "published_seconds": {"GeForce GTX 550 Ti (thesis Fig. 4.4)": 39.1},
        "mlups": round(updates / min(ts) / 1e6, 1), "result_check": status, "known_answer": key}
    # and the published 2D workload (thesis Fig. 4.2: Lyapunov FMG(2, 500, 500) on [0, 20]^2, n = 4097: 21.4 s on the thesis' GPU)
    key = "2d_n4097_fmg_2_500_500_f32"
    ka = known.get(key)
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        mg = P.MultiGrid2D(ctx, [4097] * 2, [0, 20, 0, 20], [-1, -2, 0, -3], 2, np.float32)
        mg.FullMultiGridVCycle(0, 2, 500, 500)
        got = mg.download_v(0)
        mg.close()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    s1, s2 = checksum(got)
    updates = 1000 * 2 * sum((j + 1) * (sz - 2) ** 2 for j, sz in enumerate(level_sizes(4097, 12)))
    status = "no known answer" if ka is None else ("ok" if ("%016x" % s1, "%016x" % s2) == (ka["sum64"], ka["wsum64"]) else "MISMATCH")
    out["published workload: 2D Lyapunov FMG(2,500,500) 4097^2 on [0,20]^2, 12 levels, f32, whole program"] = {
        "seconds": round(min(ts), 4), "runs_s": [round(t, 4) for t in ts], "published_seconds": {"GeForce GTX 550 Ti (thesis Fig. 4.2)": 21.4},
        "mlups": round(updates / min(ts) / 1e6, 1), "result_check": status, "known_answer": key}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=150,
                    help="untimed cycles first; the default covers the GPU's ramp-up: at 513^3 the first ~100 cycles (0.3 s) run 2-3 %% slower "
                         "than the steady state (batches with --warmup 3: 3.27, 3.25, 3.26, 3.22, 3.19 ms; with 100: 3.165 +- 0.003)")
    ap.add_argument("--size", dest="n", type=int, default=0, help="points per axis of the finest grid (2^k+1); 0 = 513 (N=1) / 1025 (N>1)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--v1", type=int, default=2)
    ap.add_argument("--v2", type=int, default=2)
    ap.add_argument("--smoother-sweeps", type=int, default=20, help="sweeps in the smoother-only roofline region")
    ap.add_argument("--min-planes", type=int, default=32,
                    help="N>1: a level stays distributed while every GPU owns this many planes; coarser levels are replicated "
                         "(below about 32 planes per GPU the ghost exchanges are pure latency)")
    ap.add_argument("--batches", type=int, default=5, help="the K timed steps are repeated this many times; the median batch is reported")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true", help="skip the result check against the committed known answer")
    ap.add_argument("--no-one-gpu-leg", action="store_true", help="N>1: skip rank 0's single-GPU run of the same problem")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    # MGX_BENCH_FORCE_DIST=1 runs the slab-decomposed code path (rendezvous, RCCL communicator, slab hierarchy) even
    # with one rank: a plumbing check for boxes with a single GPU
    force_dist = os.environ.get("MGX_BENCH_FORCE_DIST", "0") == "1"
    slabbed = world > 1 or force_dist
    rdzv = None
    if slabbed:
        rdzv = load_launch().Rendezvous(rank, world)  # control plane: no PyTorch anywhere in a rank

    import numpy as np

    import pde_multigrid_amd as P

    dtype = np.float64 if args.dtype == "f64" else np.float32
    wbytes = np.dtype(dtype).itemsize
    n = args.n or (513 if world == 1 else 1025)
    ctx = P.Context(local_rank)
    # A/B runs: MGX_BENCH_PARAMS="name=value,..." sets context parameters (speed knobs only: results never change) and is
    # reported in config.params
    params = [a.split("=") for a in os.environ.get("MGX_BENCH_PARAMS", "").split(",") if a]
    for k, v in params:
        ctx.set_param(k, int(v))

    uid = None
    if slabbed:
        uid = rdzv.broadcast_bytes(P.Context.unique_id() if rank == 0 else b"", src=0)
        ctx.comm_init(uid, rank, world)

    def make_mg(schedule):
        """the hierarchy; N > 1: `schedule` = "default" (library defaults: communication-avoiding exchanges on thick slabs, small
        levels inline on the compute stream) or "conservative" (every exchange on the comm stream, one per colour pass: the
        schedule with no collective on the compute stream and the smallest messages)"""
        if not slabbed:
            return P.MultiGrid3D(ctx, [n] * 3, R3, dtype)
        if schedule == "conservative":
            return P.DistMultiGrid3D(ctx, [n] * 3, R3, dtype, min_planes=args.min_planes, inline_bytes=0, ca_min_planes=0)
        return P.DistMultiGrid3D(ctx, [n] * 3, R3, dtype, min_planes=args.min_planes)

    # N > 1: the conservative schedule runs first and its checked result is kept; the default schedule follows under a watchdog
    # (a schedule that misbehaves on a real interconnect must not cost the run its number)
    two_legs = slabbed and world > 1 and os.environ.get("MGX_BENCH_ONE_LEG", "0") != "1"
    mg = make_mg("conservative" if two_legs else "default")
    reset = (lambda: mg.zero_v(0)) if slabbed else (lambda: mg.setToValue_v(0, 0.0, True))  # noqa: E731
    nd = mg.numDist if slabbed else 0
    nlev = mg.numGrids
    sizes = level_sizes(n, nlev)
    lups_per_cycle = (args.v1 + args.v2) * sum((s - 2) ** 3 for s in sizes)

    def barrier():
        ctx.sync()
        if rdzv is not None:
            rdzv.barrier()

    def timed_batches(cycle, steps, batches):
        """seconds of `steps` cycles, `batches` times: each batch between barrier + sync on both sides, max over ranks"""
        out = []
        for _ in range(batches):
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                cycle()
            ctx.sync()
            dt = time.perf_counter() - t0
            out.append(rdzv.max(dt) if rdzv is not None else dt)
        barrier()
        return out

    # ---- V-cycle throughput -----------------------------------------------------------
    def throughput():
        reset()
        for _ in range(args.warmup):
            mg.VCycle(0, args.v1, args.v2)
        ex0 = mg.n_exchanges if slabbed else 0
        mg.VCycle(0, args.v1, args.v2)
        exch = (mg.n_exchanges - ex0) if slabbed else None  # halo exchanges + collectives this rank enqueues per cycle
        bs = timed_batches(lambda: mg.VCycle(0, args.v1, args.v2), args.steps, max(1, args.batches))
        el = sorted(bs)[len(bs) // 2]  # the median batch
        return bs, el, lups_per_cycle * args.steps / el / 1e6, exch  # one shared problem: whole-job rate

    batch_s, elapsed, mlups, exchanges = throughput()

    # ---- smoother-only region for the roofline (HIP events on the compute stream) -------
    reset()
    e0, e1 = ctx.event(), ctx.event()
    if not slabbed:
        # the smoother exactly as the cycle calls it: Relax(grid, v1) = v1 red-black sweeps per call
        mg.Relax(0, args.v1)
        ctx.sync()
        ctx.record(e0)
        for _ in range(args.smoother_sweeps // max(args.v1, 1)):
            mg.Relax(0, args.v1)
        ctx.record(e1)
        args.smoother_sweeps = (args.smoother_sweeps // max(args.v1, 1)) * max(args.v1, 1)
        my_lups_per_launch = (n - 2) ** 3 / 2.0
        where = "finest level, x-split layout, one colour per launch"
    else:
        for c in (0, 1):
            mg.relax_colour_local(0, c)
        ctx.sync()
        ctx.record(e0)
        for _ in range(args.smoother_sweeps):
            for c in (0, 1):
                mg.relax_colour_local(0, c)
        ctx.record(e1)
        p = mg.plan(0)
        my_lups_per_launch = (n - 2) ** 2 * (p.uend - p.ubeg) / 2.0
        where = "finest level, rank 0's z-slab, one colour per launch, ghost exchange excluded"
    kname = ctx.last_relax_kernel()
    ms = ctx.elapsed_ms(e0, e1)
    launches = 2 * args.smoother_sweeps  # one launch per colour
    bytes_per_launch = 3 * wbytes * my_lups_per_launch  # 24 B/LUP fp64 per red+black sweep, half the points per colour launch
    launch_s = ms * 1e-3 / launches
    achieved = bytes_per_launch / launch_s
    smoother_mlups = 2 * my_lups_per_launch * args.smoother_sweeps / (ms * 1e-3) / 1e6
    barrier()

    # ---- the second kernel of the finest level: last black pass + residual + restrict in one launch (HIP events) -------
    down = None
    if not slabbed:
        import ctypes as C
        sfx = "f64" if args.dtype == "f64" else "f32"
        g0, g1 = mg.grid(0), mg.grid(1) if nlev > 1 else None
        n0 = (C.c_int * 3)(*g0.sizeXYZ)
        n1 = (C.c_int * 3)(*g1.sizeXYZ) if g1 is not None else None
        if g1 is not None and getattr(P.lib, "mgx3dxs_relax_rr_takes_" + sfx)(ctx._h, n0, n1):
            ct = C.c_double if args.dtype == "f64" else C.c_float
            hh = (ct * 3)(g0.h_x, g0.h_y, g0.h_z)
            fn = getattr(P.lib, "mgx3dxs_relax_rr_slab_" + sfx)

            def fused():
                P.check(fn(ctx._h, C.c_void_p(g0.d_v), C.c_void_p(g0.d_f), n0, C.c_int(0), hh, C.c_int(0), C.c_void_p(g1.d_f), n1,
                           C.c_int(0), C.c_int(1), C.c_int(g1.sizeXYZ[2] - 1)))
            fused()
            ctx.sync()
            reps = 10
            ctx.record(e0)
            for _ in range(reps):
                fused()
            ctx.record(e1)
            dms = ctx.elapsed_ms(e0, e1) / reps  # includes the zero fill of the coarse planes (1 / 8 word per point)
            nc = g1.sizeXYZ[0]
            dbytes = wbytes * (2.0 * (n - 2) ** 3 + (nc - 2) ** 3)  # red half of v + f in, black half + coarse out
            down = {"kernel": ctx.last_rr_kernel(), "what": "last black pass of the pre-smoothing + residual + restrict, finest level, one launch (timed with the zero fill of the coarse planes the stand-alone entry does first)",
                    "avg_launch_us": round(dms * 1e3, 2), "algorithmic_bytes_per_launch": dbytes,
                    "achieved": round(dbytes / (dms * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(dbytes / (dms * 1e-3) / HBM_PEAK_BPS, 4)}
    barrier()

    # ---- result check: one cycle from v = 0 against the oracle's committed known answer ----
    def result_check():
        """(check, failed): check on rank 0 only; failed on every rank"""
        check = None
        if not args.no_check and args.v1 == 2 and args.v2 == 2:
            ka = known_answer(n, nlev, args.dtype)
            reset()
            mg.VCycle(0, 2, 2)
            centre = None
            if not slabbed:
                got = mg.download_v(0)
                s1, s2 = checksum(got)
                centre = float(got[n // 2, n // 2, n // 2])
            else:
                # every rank sums the planes it owns (word index = position in the whole array); rank 0 adds the parts
                pl = mg.plan(0)
                got = mg.download_owned(0)
                s1, s2 = checksum(got, pl.zlo * n * n)
                if pl.zlo <= n // 2 < pl.zhi:
                    centre = float(got[n // 2 - pl.zlo, n // 2, n // 2])
                if world > 1:
                    parts = rdzv.all_gather((s1, s2, centre))
                    s1 = sum(q[0] for q in parts) & ((1 << 64) - 1)
                    s2 = sum(q[1] for q in parts) & ((1 << 64) - 1)
                    centre = [q[2] for q in parts if q[2] is not None][0]
            del got
            if rank == 0:
                s1, s2 = "%016x" % s1, "%016x" % s2
                if ka is None:
                    check = {"status": "no known answer committed for this size / dtype", "n": n, "dtype": args.dtype}
                else:
                    ok = s1 == ka["sum64"] and s2 == ka["wsum64"]
                    check = {"status": "ok" if ok else "MISMATCH",
                             "known_answer": "tests/golden/known_answers_f64.json: 3d_n%d_vcycle22_%dlev_%s" % (n, nlev, args.dtype),
                             "sum64": s1, "wsum64": s2, "centre": centre}
                    if not ok:
                        sys.stderr.write("bench.py: RESULT CHECK FAILED -- the cycle's result differs from the oracle's known answer: %s "
                                         "(expected sum64 %s wsum64 %s centre %r)\n"
                                         % (json.dumps(check), ka["sum64"], ka["wsum64"], ka["centre"]))
        failed = check is not None and check.get("status") == "MISMATCH"
        if rdzv is not None:
            failed = rdzv.broadcast(failed, src=0)
        return check, failed

    check, failed = result_check()
    if failed:
        sys.stderr.write("bench.py: no metric is reported\n")
        mg.close()
        ctx.close()
        if rdzv is not None:
            rdzv.close()
        sys.exit(3)

    # ---- N > 1: the same problem on ONE GPU (rank 0, the others wait), for the strong-scaling factor ----
    one_gpu = None
    if world > 1 and not args.no_one_gpu_leg:
        if rank == 0:
            ctx1 = P.Context(local_rank)
            mg1 = P.MultiGrid3D(ctx1, [n] * 3, R3, dtype)
            k1 = max(2, min(args.steps, 5))
            mg1.VCycle(0, args.v1, args.v2)
            ctx1.sync()
            ta = time.perf_counter()
            for _ in range(k1):
                mg1.VCycle(0, args.v1, args.v2)
            ctx1.sync()
            tb = time.perf_counter()
            mg1.close()
            ctx1.close()
            one_gpu = {"ms_per_step": round((tb - ta) / k1 * 1e3, 4), "steps": k1,
                       "note": "the same %d^3 hierarchy on rank 0's GPU alone, timed in this run while the other ranks wait" % n}
        rdzv.barrier()

    # ---- secondary configurations (N = 1, default workload only) ----
    secondary = None
    if world == 1 and not slabbed and not args.no_secondary and n == 513 and args.dtype == "f64" and args.v1 == 2 and args.v2 == 2:
        secondary = run_secondary(P, ctx, np)
        if any(c["result_check"] == "MISMATCH" for c in secondary.values()):
            sys.stderr.write("bench.py: RESULT CHECK FAILED in a secondary configuration: %s; no metric is reported\n" % json.dumps(secondary))
            mg.close()
            ctx.close()
            sys.exit(3)

    comm = None
    if slabbed:
        seen, ver = ctx.comm_info()
        launcher = ("bench.py's own children" if "MGX_RDZV_KEY" in os.environ else
                    "an external launcher's environment (RANK / WORLD_SIZE)" if "WORLD_SIZE" in os.environ else "this process alone")
        comm = {"ranks_seen": seen, "rccl_version": ver, "launcher": launcher, "control_plane": "pde_multigrid_amd/launch.py (TCP on 127.0.0.1)",
                "torch_imported": "torch" in sys.modules}

    def build_out(batch_s, elapsed, mlups, check, exchanges, one_gpu, schedule):
        """rank 0: the JSON line of one measured schedule"""
        if rank != 0:
            return None
        out = {
            "metric": "MLUPS on 3D Poisson 512^3 V-cycle" if n == 513 else "MLUPS on 3D Poisson %d^3 V-cycle" % (n - 1),
            "value": round(mlups, 1),
            "unit": "MLUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "batches": {"count": len(batch_s), "ms_per_step": [round(b / args.steps * 1e3, 4) for b in batch_s],
                        "min": round(min(batch_s) / args.steps * 1e3, 4), "median": round(elapsed / args.steps * 1e3, 4)},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (analytic RHS of the reference: f = -3 pi^2 sin(pi x) sin(pi y) sin(pi z), v = 0)",
            "config": {
                "workload": "3D Poisson %d^3 points (%d^3 cells), %s, V(%d,%d) cycle, %d levels%s"
                            % (n, n - 1, args.dtype, args.v1, args.v2, nlev,
                               "" if not slabbed else ", one hierarchy in %d z-slab%s" % (world, "s" if world > 1 else "")),
                "levels": sizes,
                "lups_per_cycle": lups_per_cycle,
                "parallelism": "single GPU" if not slabbed else
                               "z-slab decomposition over %d GPU%s: %d distributed levels (ghost planes over RCCL), %d replicated"
                               % (world, "s" if world > 1 else "", nd, nlev - nd),
                "note": "N=1 runs BASELINE configs[3] (513^3); N>1 runs configs[4] (1025^3) as one strong-scaled problem",
                "params": {k: int(v) for k, v in params} or None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "%s (%s)" % (kname, where),
                "achieved": round(achieved / 1e9, 1),
                "peak": HBM_PEAK_BPS / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_BPS, 4),
                "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_us": round(launch_s * 1e6, 2),
                "smoother_mlups_this_gpu": round(smoother_mlups, 1),
            },
            "result_check": check,
        }
        if down is not None:
            out["roofline_down"] = down
        if one_gpu is not None:
            one_gpu = dict(one_gpu, speedup=round(one_gpu["ms_per_step"] / (elapsed / args.steps * 1e3), 3))
            out["config"]["one_gpu_same_problem"] = one_gpu
        if slabbed:
            out["config"]["schedule"] = schedule
            out["config"]["exchanges_per_cycle"] = exchanges  # halo exchanges + collectives one rank enqueues per V-cycle
        if comm is not None:
            out["config"]["communicator"] = comm
        if secondary is not None:
            out["secondary"] = secondary
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if not slabbed and n == 513 and args.dtype == "f64" and os.path.exists(pmc):
            # HBM bytes per launch from the separate rocprofv3 --pmc passes (tools/pmc_summary.py): 2 x FETCH_SIZE +
            # WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md -- only if they were taken from the kernel that ran
            with open(pmc) as fh:
                t = json.load(fh)
            rr = t.get("relax_rr3d_f64_513")
            if down is not None and rr and rr.get("kernel", "").startswith(down["kernel"]):
                out["roofline_down"]["traffic"] = rr.get("bytes_per_launch")
            if t.get("kernel_name") == kname:
                out["roofline"]["traffic"] = t.get("smoother_f64_513_bytes_per_launch")
            else:
                out["roofline"]["traffic_note"] = ("profiles/pmc_traffic.json was taken from %r, not from the kernel that ran"
                                                   % t.get("kernel_name"))
        if world == 1 and not slabbed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, dtype)
        return out

    out = build_out(batch_s, elapsed, mlups, check, exchanges, one_gpu, "conservative" if two_legs else "default")
    if two_legs:
        # ---- the default schedule, under a watchdog: if it does not come back, rank 0 reports the conservative leg ----
        import threading
        legA = {"ms_per_step": round(elapsed / args.steps * 1e3, 4), "value": round(mlups, 1), "exchanges_per_cycle": exchanges,
                "result_check": (check or {}).get("status"), "what": "every exchange on the comm stream, one per colour pass (inline_bytes = 0, ca_min_planes = 0)"}
        deadline = float(os.environ.get("MGX_BENCH_LEG_TIMEOUT", "0")) or max(180.0, 30.0 * (time.perf_counter() - t_start))

        def give_up():
            if rank == 0:
                out["config"]["schedules"] = {"conservative": legA, "default": {"status": "did not finish within %.0f s: the conservative leg is reported" % deadline}}
                print(json.dumps(out), flush=True)
            os._exit(0)

        dog = threading.Timer(deadline, give_up)
        dog.daemon = True
        mg.close()
        dog.start()
        mg = make_mg("default")
        nd = mg.numDist
        batch_s, elapsed, mlups, exchanges = throughput()
        check, failed = result_check()
        dog.cancel()
        legB = {"ms_per_step": round(elapsed / args.steps * 1e3, 4), "value": round(mlups, 1), "exchanges_per_cycle": exchanges,
                "result_check": (check or {}).get("status"),
                "what": "library defaults: communication-avoiding exchanges (one per Relax call) on slabs of >= 16 planes, levels of <= 96 MB per slab inline on the compute stream"}
        if rank == 0:
            if failed:
                legB["status"] = "RESULT CHECK FAILED: the conservative leg is reported"
            else:
                out = build_out(batch_s, elapsed, mlups, check, exchanges, one_gpu, "default")
            out["config"]["schedules"] = {"conservative": legA, "default": legB}
    if rank == 0:
        print(json.dumps(out))
    mg.close()
    ctx.close()
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


if __name__ == "__main__":
    main()
