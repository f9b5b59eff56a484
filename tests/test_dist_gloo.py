"""CPU suite, world_size 2 over gloo: the z-slab decomposition plan of the multi-GPU V-cycle
(csrc/host/mg_dist3d.inc) emulated with the oracle's operators.

Each rank keeps whole-grid arrays but may only READ the planes of its slab window
[zoff, zoff+nzl) -- everything else is poisoned with NaN before every operator -- and only KEEPS
the planes it owns.  Ownership, ghost depths (2 below, 1 above), the exchange schedule (adjacent ghosts after
every colour pass and after the correction, the second lower ghost right before the residual, f's lower
ghost after restriction) and the agglomeration level
come from the product's own plan functions (mg_slab_plan / mg_dist_num_levels in libmgx, pure host
code).  The assembled result must equal the single-domain oracle bit for bit; a wrong ghost depth or a
missing exchange shows up as NaN or as a differing bit pattern.  The real C driver with the HIP kernels
is checked on the GPU by tests/test_gpu_dist.py (in-process transport).  The ranks are separate
processes (tests/dist_gloo_worker.py) so that this process never loads torch next to libmgx."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P

R3 = [0, 1, 0, 1, 0, 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n,min_planes,mode", [(17, 2, O.REF_COMPAT), (33, 4, O.REF_COMPAT), (33, 2, O.CORRECT)])
def test_slab_plan_world2_matches_single_domain(tmp_path, n, min_planes, mode):
    world, v1, v2, cycles = 2, 2, 2, 2
    assert P.dist_num_levels(n, world, O.num_grids(n), min_planes) >= 1
    out = str(tmp_path / "slab%d.npy")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker] + [str(a) for a in (r, world, port, n, v1, v2, cycles, min_planes, mode, out)])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.concatenate([np.load(out % r) for r in range(world)], axis=0)
    want = O.cycle3d([n] * 3, R3, mode=0, v1=v1, v2=v2, reps=cycles, residual_mode=mode, dtype=np.float64)
    assert got.shape == want.shape
    assert not np.isnan(got).any()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("n,min_planes,mode", [(33, 4, O.REF_COMPAT), (33, 2, O.CORRECT)])
def test_slab_fmg_world2_matches_single_domain(tmp_path, n, min_planes, mode):
    """FullMultiGridVCycle on slabs: Restrict(f) with the ghost below, the replicated tail's f by all-gather (its top
    boundary plane from the last rank), plain Interpolate + ghost exchange on the way up"""
    world, v0, v1, v2 = 2, 1, 2, 2
    out = str(tmp_path / "slab%d.npy")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gloo_worker.py")
    procs = [subprocess.Popen([sys.executable, worker] + [str(a) for a in (r, world, port, n, v1, v2, 0, min_planes, mode, out, v0)])
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    got = np.concatenate([np.load(out % r) for r in range(world)], axis=0)
    want = O.cycle3d([n] * 3, R3, mode=1, v0=v0, v1=v1, v2=v2, residual_mode=mode, dtype=np.float64)
    assert not np.isnan(got).any()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))


def test_plan_invariants():
    for world in (1, 2, 4, 8):
        for n in (33, 65, 513, 1025):
            ng = O.num_grids(n)
            nd = P.dist_num_levels(n, world, ng, 4)
            size = n
            for l in range(nd):
                plans = [P.slab_plan(size, r, world) for r in range(world)]
                assert plans[0].zlo == 0 and plans[-1].zhi == size
                for a, b in zip(plans, plans[1:]):
                    assert a.zhi == b.zlo            # contiguous ownership
                    assert b.zlo % 2 == 0            # owner of coarse k owns fine 2k, 2k+1
                    assert b.glo == 2 and a.ghi == 1
                for p in plans:
                    assert p.zoff == p.zlo - p.glo and p.zoff % 2 == 0
                    assert p.nzl == p.zhi - p.zlo + p.glo + p.ghi
                    assert 1 <= p.ubeg <= p.uend <= size - 1
                size = (size - 1) // 2 + 1
            assert 0 <= nd < ng or world == 1
