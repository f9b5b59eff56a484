"""Worker process of tests/test_dist_gloo.py::test_ca_*: one rank of the world-2 gloo emulation of the
COMMUNICATION-AVOIDING slab schedule (csrc/host/mg_dist3d.inc, ca_*), restated in Python with the oracle's operators.
Run as: python tests/dist_gloo_ca_worker.py rank world port n v1 v2 cycles out_pattern

Every rank keeps whole-grid arrays and, per level, plane and colour, a VERSION: how many colour passes that half-plane has
seen, or -1 for "not mine to know" (outside the slab window, or a ghost plane nobody refreshed).  A colour pass at plane
z asserts that the other colour at z - 1, z, z + 1 carries exactly the version the serial algorithm would read there --
a ghost plane trusted one pass too long, an edge / interior range off by one plane or a missing exchange trips the
assertion (and changes bits of the assembled result).  Ranges, depths and the order edges -> exchange -> interior are the
C driver's; plan and level rule come from libmgx (mg_slab_plan).  torch is imported BEFORE libmgx (see _lib.py)."""
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
D = 6            # MG_DEEP_GHOSTS
CA_MIN = 16      # mgDistMultiGrid3D::ca_min_planes
ALL = 99


def main(rank, world, port, n, v1, v2, cycles, out_path):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    dtype = np.float64
    numGrids = O.num_grids(n)
    sizes = [n]
    for _ in range(numGrids - 1):
        sizes.append((sizes[-1] - 1) // 2 + 1)
    ndist = 0
    while ndist < numGrids and (sizes[ndist] - 1) // world >= CA_MIN and ((sizes[ndist] - 1) // world) % 2 == 0:
        ndist += 1  # every distributed level of this emulation runs the CA schedule
    assert 1 <= ndist < numGrids
    plans = [P.slab_plan(sizes[l], rank, world) for l in range(ndist)]
    has_lo, has_up = rank > 0, rank < world - 1
    v = [np.zeros((s, s, s), dtype) for s in sizes[:ndist]]
    f = [np.full((s, s, s), np.nan, dtype) for s in sizes[:ndist]]
    f[0] = O.init3d([n] * 3, R3, 0, dtype)[1]
    # versions: vver[l][z, c] colour passes seen (-1 unknown), fver[l][z] 0 / -1; epoch[l][c]: the version a current plane has
    vver = [np.full((s, 2), -1, np.int64) for s in sizes[:ndist]]
    fver = [np.full(s, -1, np.int64) for s in sizes[:ndist]]
    epoch = [[0, 0] for _ in range(ndist)]
    gv, gf = [ALL] * ndist, [ALL] * ndist
    nex = [0]
    for l in range(ndist):
        p = plans[l]
        vver[l][p.zoff:p.zoff + p.nzl] = 0
        if l == 0:
            fver[l][p.zoff:p.zoff + p.nzl] = 0

    def window_only(l):  # what lies outside the slab window is unknown to this rank
        p = plans[l]
        for a in (v[l], f[l]):
            a[:p.zoff] = np.nan
            a[p.zoff + p.nzl:] = np.nan

    def exchange(l, which, d):  # ca_exchange_: d planes either way, values and versions
        p = plans[l]
        arr, ver = (f[l], fver[l]) if which else (v[l], vver[l])
        assert 1 <= d <= D and d <= (sizes[l] - 1) // world
        nex[0] += 1
        reqs, bufs = [], {}
        if has_lo:
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(arr[p.zlo:p.zlo + d])), rank - 1, tag=1))
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(ver[p.zlo:p.zlo + d]).astype(np.int64)), rank - 1, tag=2))
            bufs["lo"] = (np.empty((d,) + arr.shape[1:], dtype), np.empty((d,) + ver.shape[1:], np.int64))
            reqs.append(dist.irecv(torch.from_numpy(bufs["lo"][0]), rank - 1, tag=3))
            reqs.append(dist.irecv(torch.from_numpy(bufs["lo"][1]), rank - 1, tag=4))
        if has_up:
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(arr[p.zhi - d:p.zhi])), rank + 1, tag=3))
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(ver[p.zhi - d:p.zhi]).astype(np.int64)), rank + 1, tag=4))
            bufs["up"] = (np.empty((d,) + arr.shape[1:], dtype), np.empty((d,) + ver.shape[1:], np.int64))
            reqs.append(dist.irecv(torch.from_numpy(bufs["up"][0]), rank + 1, tag=1))
            reqs.append(dist.irecv(torch.from_numpy(bufs["up"][1]), rank + 1, tag=2))
        for r in reqs:
            r.wait()
        if has_lo:
            arr[p.zlo - d:p.zlo], ver[p.zlo - d:p.zlo] = bufs["lo"]
        if has_up:
            arr[p.zhi:p.zhi + d], ver[p.zhi:p.zhi + d] = bufs["up"]

    def need(l, which, d):  # ca_need_
        st = gf if which else gv
        if d > 0 and st[l] < d:
            exchange(l, which, d)
            st[l] = d

    fault = int(os.environ.get("CA_EMU_FAULT", "0"))  # negative test: 1 = every pass trusts one ghost plane more than it may

    def colour_pass(l, j, zero0, za, zb):  # ca_pass_ on the global planes [za, zb): pass j, colour j & 1
        N = sizes[l] - 1
        if fault == 1 and has_lo and za < plans[l].zlo:
            za -= 1
        za, zb = max(za, 1), min(zb, N)
        if zb <= za:
            return
        c = j & 1
        s3 = [sizes[l]] * 3
        want_other = epoch[l][1 - c] + (j + c) // 2  # passes of the other colour before pass j: j // 2 (+ 1 if this one is black)
        for z in range(za, zb):
            assert fver[l][z] == 0, ("f unknown", l, j, z)
            for zz in (z - 1, z, z + 1):
                if 0 < zz < N and not zero0:
                    assert vver[l][zz, 1 - c] == want_other, ("stale neighbour", rank, l, j, z, zz, int(vver[l][zz, 1 - c]), want_other)
        src = np.zeros_like(v[l]) if zero0 else v[l]
        t = O.relax_colour3d(s3, R3, np.nan_to_num(src, nan=1e300), np.nan_to_num(f[l], nan=1e300), c, dtype)
        zi, yi, xi = np.nonzero(np.ones((zb - za,) + v[l].shape[1:], bool))
        m = ((xi + yi + zi + za) & 1) == c
        sel = (zi[m] + za, yi[m], xi[m])
        inner = (sel[1] > 0) & (sel[1] < sizes[l] - 1) & (sel[2] > 0) & (sel[2] < sizes[l] - 1)
        sel = tuple(a[inner] for a in sel)
        v[l][sel] = t[sel]
        vver[l][za:zb, c] = epoch[l][c] + j // 2 + 1

    def can_split(l, k, x):
        return world > 1 and x > 0 and (sizes[l] - 1) // world >= 2 * x + 2 * (k - 1) + 3

    def relax(l, K, zero, x):  # ca_relax_ with o = 0, kint = K
        p = plans[l]
        N = sizes[l] - 1
        if K <= 0:
            need(l, 0, x)
            return
        j0 = 0
        while j0 < K:
            if K - j0 <= D:
                k, xc = K - j0, x
            else:
                k = min(K - j0, D)
                xc = min(K - j0 - k, D) if K - j0 - k > 0 else x
            zero0 = zero and j0 == 0
            if zero0:  # all zeros by contract
                gv[l] = ALL
                v[l][:] = 0
                vver[l][p.zoff:p.zoff + p.nzl] = np.array(epoch[l])
            need(l, 1, k - 1)
            need(l, 0, k)
            gv[l] = 0
            if can_split(l, k, xc):
                for jj in range(k):  # the edges first
                    r = k - 1 - jj
                    if has_lo:
                        colour_pass(l, j0 + jj, zero0 and jj == 0 and j0 == 0, p.zlo - r, p.zlo + xc + r)
                    if has_up:
                        colour_pass(l, j0 + jj, zero0 and jj == 0 and j0 == 0, p.zhi - xc - r, p.zhi + r)
                exchange(l, 0, xc)
                for jj in range(k):
                    r = k - 1 - jj
                    colour_pass(l, j0 + jj, zero0 and jj == 0 and j0 == 0, p.zlo + xc + r if has_lo else 1, p.zhi - xc - r if has_up else N)
                gv[l] = xc
            else:
                for jj in range(k):
                    r = k - 1 - jj
                    colour_pass(l, j0 + jj, zero0 and jj == 0 and j0 == 0, p.zlo - r if has_lo else 1, p.zhi + r if has_up else N)
                gv[l] = 0
                if xc > 0:
                    exchange(l, 0, xc)
                    gv[l] = xc
            j0 += k
        epoch[l][0] += K // 2
        epoch[l][1] += K // 2
        # what did not keep up is unknown from now on
        stale = (vver[l][:, 0] != epoch[l][0]) | (vver[l][:, 1] != epoch[l][1])
        stale[0] = stale[-1] = False
        vver[l][stale] = -1
        v[l][stale] = np.nan

    def first_chunk(sweeps):
        return min(2 * sweeps, D)

    def x_post(l):
        return max(first_chunk(v1), 2) if l == 0 else (first_chunk(v2) // 2 + 1 if v2 > 0 else 1)

    def current(l, z):
        N = sizes[l] - 1
        return z <= 0 or z >= N or (vver[l][z, 0] == epoch[l][0] and vver[l][z, 1] == epoch[l][1])

    def vcycle(l, v_zero):
        p = plans[l]
        s3 = [sizes[l]] * 3
        N = sizes[l] - 1
        cN = sizes[l + 1] - 1
        czlo = p.zlo // 2
        # ---- ca_down_
        zero = False
        if v_zero:
            if v1 > 0:
                zero = True
            else:
                v[l][:] = 0
                vver[l][p.zoff:p.zoff + p.nzl] = np.array(epoch[l])
                gv[l] = ALL
        x = max(first_chunk(v2) if v2 > 0 else 2, 2)
        relax(l, 2 * v1, zero, x)
        if gf[l] < 1:
            need(l, 1, 1)
        need(l, 0, 2)
        rr_end = (sizes[l + 1] if rank == world - 1 else p.zhi // 2) if l + 1 < ndist else czlo + cN // world
        for pz in range(max(czlo, 1), min(rr_end, cN)):  # what residual + restrict of my coarse planes read
            for z in range(2 * pz - 2, 2 * pz + 3):
                assert current(l, z), ("residual reads a stale plane", rank, l, pz, z)
            for z in range(2 * pz - 1, 2 * pz + 2):
                assert fver[l][z] == 0 or z <= 0 or z >= N
        window_only(l)
        r = O.residual3d(s3, R3, np.nan_to_num(v[l], nan=1e300), np.nan_to_num(f[l], nan=1e300), O.REF_COMPAT, dtype)
        cf = O.restrict3d(s3, r, dtype)
        if l + 1 < ndist:
            f[l + 1][:] = np.nan
            fver[l + 1][:] = -1
            f[l + 1][czlo:rr_end] = cf[czlo:rr_end]
            fver[l + 1][czlo:rr_end] = 0
            gf[l + 1] = 0
            vcycle(l + 1, True)
            # ---- ca_up_: the coarse planes under [zlo - k0, zhi + k0)
            k0 = first_chunk(v2)
            need(l, 0, k0)
            need(l + 1, 0, k0 // 2 + 1)
            cv = v[l + 1].copy()
            cver = vver[l + 1]
            cep = epoch[l + 1]
            zmin, zmax = (p.zlo - k0 if has_lo else 1), (p.zhi + k0 if has_up else N)
            for pz in range(zmin // 2, zmax // 2 + 1):
                assert pz <= 0 or pz >= cN or (cver[pz, 0] == cep[0] and cver[pz, 1] == cep[1]), ("stale coarse plane", rank, l, pz)
        else:
            share = cN // world
            mine = torch.from_numpy(np.ascontiguousarray(cf[czlo:czlo + share]))
            parts = [torch.empty_like(mine) for _ in range(world)]
            nex[0] += 1
            dist.all_gather(parts, mine)
            tf = np.zeros((sizes[l + 1],) * 3, dtype)
            tf[:cN] = np.concatenate([q.numpy() for q in parts], axis=0)
            cv = O.cycle3d([sizes[l + 1]] * 3, R3, nlevels=numGrids - ndist, mode=0, v1=v1, v2=v2, v=np.zeros_like(tf), f=tf,
                           residual_mode=O.REF_COMPAT, dtype=dtype)
            k0 = first_chunk(v2)
            need(l, 0, k0)
            zmin, zmax = (p.zlo - k0 if has_lo else 1), (p.zhi + k0 if has_up else N)
        assert v2 > 0
        for z in range(max(zmin, 1), zmax):
            assert current(l, z), ("the correction meets a stale plane", rank, l, z)
        e = O.interpolate3d(s3, np.zeros_like(v[l]), np.nan_to_num(cv, nan=1e300), dtype)
        lo, hi = max(2 * (zmin // 2), 1), 2 * (zmax // 2)
        v[l][lo:hi] = v[l][lo:hi] + e[lo:hi]  # every point; the driver corrects the black ones only (the red pass rewrites red)
        relax(l, 2 * v2, False, x_post(l))

    per_cycle = []
    for _ in range(cycles):
        before = nex[0]
        vcycle(0, False)
        per_cycle.append(nex[0] - before)
    p = plans[0]
    own = v[0][p.zlo:p.zhi]
    assert not np.isnan(own).any()
    np.save(out_path % rank, own)
    np.save((out_path % rank) + ".count.npy", np.array(per_cycle))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), a[7])
