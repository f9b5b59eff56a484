"""Worker process of tests/test_dist_gloo.py (one rank of the world-2 gloo emulation of the z-slab plan).
Run as: python tests/dist_gloo_worker.py rank world port n v1 v2 cycles min_planes mode out_pattern
torch is imported BEFORE libmgx on purpose: torch bundles its own ROCm runtime libraries and the two
must not be loaded in the other order in one process (see pde_multigrid_amd/_lib.py)."""
import os
import sys

import torch  # noqa: F401  (first, see above)
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import oracle as O  # noqa: E402
import pde_multigrid_amd as P  # noqa: E402

R3 = [0, 1, 0, 1, 0, 1]


def _rank_main(rank, world, port, n, v1, v2, cycles, min_planes, mode, out_path, fmg_v0=0):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dtype = np.float64
    numGrids = O.num_grids(n)
    ndist = P.dist_num_levels(n, world, numGrids, min_planes)
    sizes = [n]
    for _ in range(numGrids - 1):
        sizes.append((sizes[-1] - 1) // 2 + 1)
    plans = [P.slab_plan(sizes[l], rank, world) for l in range(ndist)]
    v = [np.zeros((s, s, s), dtype) for s in sizes[:ndist]]
    f = [np.zeros((s, s, s), dtype) for s in sizes[:ndist]]
    f[0] = O.init3d([n] * 3, R3, 0, dtype)[1]

    def poison(a, p):
        a[:p.zoff] = np.nan
        a[p.zoff + p.nzl:] = np.nan

    def sendrecv(send_lo, recv_lo, send_up, recv_up):
        reqs = []
        if rank > 0:
            if send_lo is not None:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(send_lo)), rank - 1))
            if recv_lo is not None:
                reqs.append(dist.irecv(torch.from_numpy(recv_lo), rank - 1))
        if rank < world - 1:
            if send_up is not None:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(send_up)), rank + 1))
            if recv_up is not None:
                reqs.append(dist.irecv(torch.from_numpy(recv_up), rank + 1))
        for r in reqs:
            r.wait()

    def exchange_v(l):  # 1 plane up, 1 plane down (slab_exchange_): ghosts zlo-1 and zhi
        p, a = plans[l], v[l]
        top = p.zhi - 1 if rank == world - 1 else p.zhi
        lo_buf = np.empty((1,) + a.shape[1:], dtype) if rank > 0 else None
        up_buf = np.empty((1,) + a.shape[1:], dtype) if rank < world - 1 else None
        if rank > 0:
            a[p.zlo - 2] = np.nan  # the second ghost goes stale: only exchange_v2 may make it valid again
        sendrecv(a[p.zlo:p.zlo + 1], lo_buf, a[top - 1:top], up_buf)
        if rank > 0:
            a[p.zlo - 1:p.zlo] = lo_buf
        if rank < world - 1:
            a[p.zhi:p.zhi + 1] = up_buf

    def exchange_v2(l):  # slab_exchange2_: ghost zlo-2 <- lower neighbour's plane top-2
        p, a = plans[l], v[l]
        top = p.zhi - 1 if rank == world - 1 else p.zhi
        lo_buf = np.empty((1,) + a.shape[1:], dtype) if rank > 0 else None
        sendrecv(None, lo_buf, a[top - 2:top - 1], None)
        if rank > 0:
            a[p.zlo - 2:p.zlo - 1] = lo_buf

    def exchange_f_up(l):  # slab_exchange_f_
        p, a = plans[l], f[l]
        top = p.zhi - 1 if rank == world - 1 else p.zhi
        lo_buf = np.empty((1,) + a.shape[1:], dtype) if rank > 0 else None
        sendrecv(None, lo_buf, a[top - 1:top], None)
        if rank > 0:
            a[p.zlo - 1:p.zlo] = lo_buf

    def relax(l, k):
        p = plans[l]
        s3 = [sizes[l]] * 3
        for _ in range(k):
            for colour in (0, 1):
                poison(v[l], p)
                poison(f[l], p)
                t = O.relax_colour3d(s3, R3, v[l], f[l], colour, dtype)
                v[l][p.ubeg:p.uend] = t[p.ubeg:p.uend]
                exchange_v(l)

    def vcycle(l):
        p = plans[l]
        s3 = [sizes[l]] * 3
        N = sizes[l] - 1
        relax(l, v1)
        if l != numGrids - 1:
            exchange_v2(l)
            poison(v[l], p)
            poison(f[l], p)
            r = O.residual3d(s3, R3, v[l], f[l], mode, dtype)
            cf = O.restrict3d(s3, r, dtype)
            cN = sizes[l + 1] - 1
            czlo = p.zlo // 2
            pzint = min(p.zhi, N) // 2
            if l + 1 < ndist:
                czhi = sizes[l + 1] if rank == world - 1 else p.zhi // 2
                f[l + 1][:] = np.nan
                f[l + 1][czlo:czhi] = cf[czlo:czhi]
                exchange_f_up(l + 1)
                v[l + 1][:] = 0
                vcycle(l + 1)
                cv = v[l + 1].copy()
                poison(cv, plans[l + 1])
            else:
                share = cN // world
                mine = torch.from_numpy(np.ascontiguousarray(cf[czlo:czlo + share]))
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                tf = np.zeros((sizes[l + 1],) * 3, dtype)
                tf[:cN] = np.concatenate([q.numpy() for q in parts], axis=0)
                cv = O.cycle3d([sizes[l + 1]] * 3, R3, nlevels=numGrids - ndist, mode=0, v1=v1, v2=v2,
                               v=np.zeros_like(tf), f=tf, residual_mode=mode, dtype=dtype)
            e = O.interpolate3d(s3, np.zeros_like(v[l]), cv, dtype)
            lo, hi = max(2 * czlo, 1), 2 * pzint
            v[l][lo:hi] = v[l][lo:hi] + e[lo:hi]
            exchange_v(l)
        relax(l, v2)

    def fmg(l):  # mgDistMultiGrid3D_FullMultiGridVCycle (csrc/host/mg_dist3d.inc) on a distributed level
        p = plans[l]
        s3 = [sizes[l]] * 3
        N = sizes[l] - 1
        cN = sizes[l + 1] - 1
        czlo = p.zlo // 2
        pzint = min(p.zhi, N) // 2
        poison(f[l], p)
        cf = O.restrict3d(s3, f[l], dtype)  # Restrict(f): the owned coarse planes read the f ghost below
        if l + 1 < ndist:
            czhi = sizes[l + 1] if rank == world - 1 else p.zhi // 2
            f[l + 1][:] = np.nan
            f[l + 1][czlo:czhi] = cf[czlo:czhi]
            exchange_f_up(l + 1)
            fmg(l + 1)
            cv = v[l + 1].copy()
            poison(cv, plans[l + 1])
        else:
            share = cN // world
            mine = torch.from_numpy(np.ascontiguousarray(cf[czlo:czlo + share]))
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine)
            tf = np.zeros((sizes[l + 1],) * 3, dtype)
            tf[:cN] = np.concatenate([q.numpy() for q in parts], axis=0)
            top = torch.from_numpy(np.ascontiguousarray(cf[cN:cN + 1]))  # the injected boundary plane: valid on the last rank
            tops = [torch.empty_like(top) for _ in range(world)]
            dist.all_gather(tops, top)
            tf[cN:] = tops[world - 1].numpy()
            cv = O.cycle3d([sizes[l + 1]] * 3, R3, nlevels=numGrids - ndist, mode=1, v0=fmg_v0, v1=v1, v2=v2,
                           v=np.zeros_like(tf), f=tf, residual_mode=mode, dtype=dtype)
        e = O.interpolate3d(s3, v[l], cv, dtype)  # plain Interpolate: interior of the fine planes of my coarse cells
        lo, hi = max(2 * czlo, 1), 2 * pzint
        v[l][lo:hi] = e[lo:hi]
        exchange_v(l)
        for _ in range(fmg_v0):
            vcycle(l)

    if fmg_v0:
        fmg(0)
    for _ in range(cycles):
        vcycle(0)
    p = plans[0]
    np.save(out_path % rank, v[0][p.zlo:p.zhi])
    # mgDistMultiGrid3D_ResidualNorm: squared residual over the owned interior planes (their neighbours zlo-1 / zhi are
    # fresh ghosts after the last exchange), one all-reduce (sum) of one double
    poison(v[0], p)
    poison(f[0], p)
    r = O.residual3d([n] * 3, R3, v[0], f[0], mode, dtype)
    mine = torch.tensor([float(np.sum(r[p.ubeg:p.uend] ** 2))], dtype=torch.float64)
    dist.all_reduce(mine, op=dist.ReduceOp.SUM)
    np.save((out_path % rank) + ".norm.npy", np.array([np.sqrt(mine.item())]))
    dist.barrier()
    dist.destroy_process_group()



if __name__ == "__main__":
    a = sys.argv[1:]
    _rank_main(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7]), int(a[8]), a[9],
               int(a[10]) if len(a) > 10 else 0)
