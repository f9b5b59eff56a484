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
    # the distributed residual norm (slab sums + all-reduce of one double): same value on both ranks, equal to the norm of
    # the single-domain residual up to the summation order (an addition without a reference: parity unpinned)
    norms = [float(np.load((out % r) + ".norm.npy")[0]) for r in range(world)]
    f0 = O.init3d([n] * 3, R3, 0, np.float64)[1]
    ref = float(np.sqrt(np.sum(O.residual3d([n] * 3, R3, want, f0, mode, np.float64) ** 2)))
    assert norms[0] == norms[1] and np.isfinite(norms[0])
    assert abs(norms[0] - ref) <= 1e-12 * ref


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


@pytest.mark.parametrize("n,v1,v2,cycles", [(65, 2, 2, 2), (65, 1, 3, 1), (33, 2, 1, 2), (65, 4, 2, 1)])
def test_ca_schedule_world2_matches_single_domain(tmp_path, n, v1, v2, cycles):
    """the communication-avoiding schedule (slabs of >= 16 planes: 6 ghost planes, one exchange of v per Relax call, the first
    ghost planes relaxed redundantly, edges first / interior behind the exchange) restated with the oracle's operators
    (tests/dist_gloo_ca_worker.py): every half-plane carries the number of colour passes it has seen and every pass asserts that
    it reads exactly the version the serial algorithm reads.  65^3 on 2 ranks: level 0 (32 planes per rank) splits into edges
    and interior, level 1 (16) does not; 33^3: one level; V(4,2): eight passes = two chunks with an exchange in between."""
    world = 2
    out = str(tmp_path / "ca%d.npy")
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gloo_ca_worker.py")
    procs = [subprocess.Popen([sys.executable, worker] + [str(a) for a in (r, world, port, n, v1, v2, cycles, out)]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    got = np.concatenate([np.load(out % r) for r in range(world)], axis=0)
    want = O.cycle3d([n] * 3, R3, mode=0, v1=v1, v2=v2, reps=cycles, dtype=np.float64)
    assert got.shape == want.shape and not np.isnan(got).any()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    counts = [np.load((out % r) + ".count.npy").tolist() for r in range(world)]
    assert counts[0] == counts[1]
    if (v1, v2) == (2, 2) and n == 65:  # level 0: v behind pre- and post-smoothing; level 1: f, v, v; the all-gather
        assert max(counts[0]) <= 6, counts


def test_ca_schedule_emulation_notices_a_ghost_plane_trusted_too_long(tmp_path):
    """the emulation's version check is what proves a schedule right: with every pass reaching one ghost plane further than its
    inputs allow (CA_EMU_FAULT=1) a rank must fail"""
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gloo_ca_worker.py")
    env = dict(os.environ, CA_EMU_FAULT="1")
    procs = [subprocess.Popen([sys.executable, worker] + [str(a) for a in (r, world, port, 65, 2, 2, 1, str(tmp_path / "f%d.npy"))], env=env,
                              stderr=subprocess.DEVNULL) for r in range(world)]
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:  # the peer of a failed rank waits for a message that never comes
            p.kill()
            codes.append(-9)
    assert any(c not in (0, -9) for c in codes), codes


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
                    deep = (size - 1) // world >= 8  # MG_DEEP_MIN_PLANES: 6 ghost planes either side, else 2 below / 1 above
                    assert (b.glo, a.ghi) == ((6, 6) if deep else (2, 1))
                for p in plans:
                    assert p.zoff == p.zlo - p.glo and p.zoff % 2 == 0 and p.zoff >= 0 and p.zoff + p.nzl <= size
                    assert p.nzl == p.zhi - p.zlo + p.glo + p.ghi
                    assert 1 <= p.ubeg <= p.uend <= size - 1
                size = (size - 1) // 2 + 1
            assert 0 <= nd < ng or world == 1


def test_bench_checksum_of_slabs_adds_up():
    """bench.py's result check at N > 1: every rank sums the planes it owns with the word index of the whole array and
    rank 0 adds the parts mod 2^64 -- equal to the checksum of the assembled array (and to oracle/gen_known_f64.py's)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    spec = importlib.util.spec_from_file_location("gen_known", os.path.join(root, "oracle", "gen_known_f64.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = np.random.default_rng(0)
    for dtype in (np.float64, np.float32):
        a = rng.uniform(-1, 1, (33, 17, 9)).astype(dtype)
        whole = bench.checksum(a)
        assert ("%016x" % whole[0], "%016x" % whole[1]) == gen.checksum64(a)
        for world in (2, 4, 8):
            s1 = s2 = 0
            for r in range(world):
                pl = P.slab_plan(33, r, world)
                q1, q2 = bench.checksum(a[pl.zlo:pl.zhi], pl.zlo * 17 * 9)
                s1, s2 = (s1 + q1) & ((1 << 64) - 1), (s2 + q2) & ((1 << 64) - 1)
            assert (s1, s2) == whole
