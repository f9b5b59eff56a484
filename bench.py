#!/usr/bin/env python3
"""bench.py -- MLUPS of the 3D Poisson V(2,2) cycle on MI355X, with the smoother's HBM roofline.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload
  N = 1 : BASELINE.json configs[3]: 3D Poisson, 513 points per axis ("512^3"), fp64, native 9-level
          hierarchy, analytic RHS of the reference (Grid3D::InitF), v = 0, reference (REF_COMPAT) semantics.
  N > 1 : BASELINE.json configs[4]: the same problem at 1025 points per axis ("1024^3"), ONE hierarchy
          decomposed into z-slabs over the N GPUs (ghost planes over RCCL/xGMI, coarse levels replicated after an
          all-gather) -- strong scaling: the total work does not depend on N.  `--n` overrides the size.
One step = one VCycle(0, 2, 2) through the C host layer (include/mg_multigrid.h), inputs resident in HBM.
    MLUPS = (v1+v2) * sum_levels (n_l - 2)^3 * steps / seconds          
roofline: the dominant kernel is the red-black Gauss-Seidel smoother on the finest level; its algorithmic
    traffic is 3 reals per lattice update per red+black sweep = 24 B/LUP in fp64 (12 B per LUP of one colour
    launch).  `achieved` = algorithmic bytes per launch / average launch duration, measured here with HIP events
    on the stream the kernel runs on, over a smoother-only timed region (rank 0's slab when N > 1, without the
    ghost exchange).
cpu_baseline: the oracle's CPU restatement ("port": same loop nest and single thread as the reference) timed
    on this box's host cores on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", dest="n", type=int, default=0, help="points per axis of the finest grid (2^k+1); 0 = 513 (N=1) / 1025 (N>1)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--v1", type=int, default=2)
    ap.add_argument("--v2", type=int, default=2)
    ap.add_argument("--smoother-sweeps", type=int, default=20, help="sweeps in the smoother-only roofline region")
    ap.add_argument("--min-planes", type=int, default=32,
                    help="N>1: a level stays distributed while every GPU owns this many planes; coarser levels are replicated "
                         "(below about 32 planes per GPU the ghost exchanges are pure latency)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    # MGX_BENCH_FORCE_DIST=1 runs the slab-decomposed code path (torch rendezvous, RCCL communicator, slab
    # hierarchy) even with one rank: a plumbing check for boxes with a single GPU
    force_dist = os.environ.get("MGX_BENCH_FORCE_DIST", "0") == "1"
    dist = None
    if world > 1 or force_dist:
        # torch BEFORE libmgx (pde_multigrid_amd/_lib.py: load order of the ROCm runtime libraries)
        import torch
        import torch.distributed as dist  # control plane only (rendezvous, barrier, max over ranks); data plane = RCCL in libmgx
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import numpy as np

    import pde_multigrid_amd as P

    dtype = np.float64 if args.dtype == "f64" else np.float32
    wbytes = np.dtype(dtype).itemsize
    n = args.n or (513 if world == 1 else 1025)
    ctx = P.Context(local_rank)

    if world == 1 and not force_dist:
        mg = P.MultiGrid3D(ctx, [n] * 3, R3, dtype)
        reset = lambda: mg.setToValue_v(0, 0.0, True)  # noqa: E731
        nd = 0
    else:
        uid = [P.Context.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(uid[0], rank, world)
        mg = P.DistMultiGrid3D(ctx, [n] * 3, R3, dtype, min_planes=args.min_planes)
        reset = lambda: mg.zero_v(0)  # noqa: E731
        nd = mg.numDist
    nlev = mg.numGrids
    sizes = level_sizes(n, nlev)
    lups_per_cycle = (args.v1 + args.v2) * sum((s - 2) ** 3 for s in sizes)

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    # ---- V-cycle throughput -----------------------------------------------------------
    reset()
    for _ in range(args.warmup):
        mg.VCycle(0, args.v1, args.v2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mg.VCycle(0, args.v1, args.v2)
    ctx.sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    mlups = lups_per_cycle * args.steps / elapsed / 1e6  # one shared problem: whole-job rate

    # ---- smoother-only region for the roofline (HIP events on the compute stream) -------
    reset()
    e0, e1 = ctx.event(), ctx.event()
    if world == 1 and not force_dist:
        # the smoother exactly as the cycle calls it: Relax(grid, v1) = v1 red-black sweeps per call
        mg.Relax(0, args.v1)
        ctx.sync()
        ctx.record(e0)
        for _ in range(args.smoother_sweeps // max(args.v1, 1)):
            mg.Relax(0, args.v1)
        ctx.record(e1)
        args.smoother_sweeps = (args.smoother_sweeps // max(args.v1, 1)) * max(args.v1, 1)
        my_lups_per_launch = (n - 2) ** 3 / 2.0
        kname = "relax3d_xs_pipe_kernel<%s,2,8,2> (finest level, x-split layout, one colour per launch)"
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
        kname = "relax3d_xs_pipe_kernel<%s,2,8,2> (finest level, rank 0's z-slab, one colour per launch, ghost exchange excluded)"
    ms = ctx.elapsed_ms(e0, e1)
    launches = 2 * args.smoother_sweeps  # one launch per colour
    bytes_per_launch = 3 * wbytes * my_lups_per_launch  # 24 B/LUP fp64 per red+black sweep, half the points per colour launch
    launch_s = ms * 1e-3 / launches
    achieved = bytes_per_launch / launch_s
    smoother_mlups = 2 * my_lups_per_launch * args.smoother_sweeps / (ms * 1e-3) / 1e6
    barrier()

    if rank == 0:
        out = {
            "metric": "MLUPS on 3D Poisson 512^3 V-cycle" if n == 513 else "MLUPS on 3D Poisson %d^3 V-cycle" % (n - 1),
            "value": round(mlups, 1),
            "unit": "MLUPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic (analytic RHS of the reference: f = -3 pi^2 sin(pi x) sin(pi y) sin(pi z), v = 0)",
            "config": {
                "workload": "3D Poisson %d^3 points (%d^3 cells), %s, V(%d,%d) cycle, %d levels%s"
                            % (n, n - 1, args.dtype, args.v1, args.v2, nlev,
                               "" if world == 1 else ", one hierarchy in %d z-slabs" % world),
                "levels": sizes,
                "lups_per_cycle": lups_per_cycle,
                "parallelism": "single GPU" if world == 1 else
                               "z-slab decomposition over %d GPUs: %d distributed levels (ghost planes over RCCL), %d replicated"
                               % (world, nd, nlev - nd),
                "note": "N=1 runs BASELINE configs[3] (513^3); N>1 runs configs[4] (1025^3) as one strong-scaled problem",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname % ("double" if wbytes == 8 else "float"),
                "achieved": round(achieved / 1e9, 1),
                "peak": HBM_PEAK_BPS / 1e9,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_BPS, 4),
                "traffic": None,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_us": round(launch_s * 1e6, 2),
                "smoother_mlups_this_gpu": round(smoother_mlups, 1),
            },
        }
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if world == 1 and n == 513 and args.dtype == "f64" and os.path.exists(pmc):
            # HBM bytes per launch of the same kernel from the separate rocprofv3 --pmc passes (tools/pmc_summary.py):
            # 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md
            with open(pmc) as fh:
                out["roofline"]["traffic"] = json.load(fh).get("smoother_f64_513_bytes_per_launch")
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import oracle as O  # CPU baseline leg only: the checker timed as a reported baseline
            cn, clev, reps = 257, 6, 12  # about 10 s of single-threaded CPU work
            secs = O.time_vcycle3d(cn, clev, args.v1, args.v2, reps, dtype)
            c_lups = (args.v1 + args.v2) * sum((s - 2) ** 3 for s in level_sizes(cn, clev)) * reps
            out["cpu_baseline"] = {
                "value": round(c_lups / secs / 1e6, 2),
                "unit": "MLUPS",
                "cores": 1,
                "kind": "port",
                "sample": "%d V(%d,%d) cycles, 3D Poisson %d^3 %s, %d levels, oracle CPU restatement (-O2, reference loop "
                          "nest, 1 thread of %d host cores), %.1f s" % (reps, args.v1, args.v2, cn, args.dtype, clev,
                                                                         os.cpu_count() or 0, secs),
            }
        print(json.dumps(out))
    mg.close()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
