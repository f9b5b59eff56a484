"""GPU suite: the slab-decomposed V-cycle driver (csrc/host/mg_dist3d.inc) with the real HIP kernels.
The gpurun box has one GPU, so the ranks are host threads of this process, one context each on the same
device, joined by the in-process test transport (mgx_comm_init_local).  Like RCCL it is asynchronous: an
exchange only enqueues device-to-device copies on the comm stream, ordered against the neighbours by
cross-context events, and the compute stream sees the ghosts only through mgx_comm_wait -- so the overlap
schedule's event ordering is what is being tested, not just its arithmetic.  Every test runs with the
transport's delay hook on (each transfer starts DELAY_US late on the receiving comm stream): a missing or
misplaced wait then reads stale ghosts for certain, and test_dist_missing_wait_is_detected proves it.
Everything else -- slab plan, ghost handling, z-range kernels, agglomeration through the replicated tail --
is the production code.  Bar: bit-identical to the single-GPU hierarchy and to the oracle."""
import threading

import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal

pytestmark = pytest.mark.gpu
R3 = [0, 1, 0, 1, 0, 1]
DELAY_US = 300  # a colour pass of these grids takes 5-50 us: every transfer arrives long after the next kernel could start


def run_ranks(nranks, n3, rng, dtype, v1, v2, cycles, min_planes, mode=P.REF_COMPAT, v0=None, f0=None, nlevels=0, fmg=0,
              delay_us=DELAY_US, drop_waits=False, extra=None, join_timeout=100, inline_bytes=0, v_levels=None, params=None,
              pack_halos=None, ca_min_planes=None, counts=None):
    """inline_bytes = 0: every level runs the OVERLAPPED schedule (comm stream, edge planes first) -- what these tests were
    written for; None: the library default (small levels exchange inline on the compute stream); a number: that threshold"""
    ctxs = [P.Context(0) for _ in range(nranks)]
    for c in ctxs:
        for k, val in (params or {}).items():
            c.set_param(k, val)
    group = P.LocalGroup(nranks)
    group.set_test_hooks(delay_us, drop_waits)
    for r, c in enumerate(ctxs):
        group.attach(c, r)
    full = np.full(tuple(reversed(n3)), np.nan, dtype)
    errors, info = [], {}

    def worker(r):
        try:
            mg = P.DistMultiGrid3D(ctxs[r], n3, rng, dtype, nlevels=nlevels, residual_mode=mode, min_planes=min_planes,
                                   inline_bytes=inline_bytes, pack_halos=pack_halos, ca_min_planes=ca_min_planes)
            info[r] = (mg.numDist, mg.numGrids)
            if f0 is not None:
                mg.upload_f(0, f0)
            if v0 is not None:
                mg.upload_v(0, v0)
            for lvl, arr in (v_levels or {}).items():
                mg.upload_v(lvl, arr)
            if fmg:
                mg.FullMultiGridVCycle(0, fmg, v1, v2)
            for _ in range(cycles):
                before = mg.n_exchanges
                mg.VCycle(0, v1, v2)
                if counts is not None:
                    counts.setdefault(r, []).append(mg.n_exchanges - before)  # exchanges + collectives of this cycle
            if extra is not None:
                info[("extra", r)] = extra(mg)
            mg.download_v_into(0, full)
            mg.close()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=join_timeout)
    alive = [t for t in threads if t.is_alive()]
    assert not errors, errors
    assert not alive, "a rank is stuck in an exchange (another rank failed or the schedule is unbalanced)"
    for c in ctxs:
        c.close()
    group.close()
    return full, info


@pytest.mark.timeout(150)
@pytest.mark.parametrize("nranks", [1, 2, 4])
@pytest.mark.parametrize("n,min_planes", [(33, 2), (65, 4), (65, 2)])
def test_dist_vcycle_matches_single_gpu_and_oracle(nranks, n, min_planes):
    got, info = run_ranks(nranks, [n] * 3, R3, np.float64, 2, 2, 2, min_planes)
    assert info[0][0] == P.dist_num_levels(n, nranks, O.num_grids(n), min_planes) >= 1
    want = O.cycle3d([n] * 3, R3, mode=0, v1=2, v2=2, reps=2, dtype=np.float64)
    assert not np.isnan(got).any()
    assert bits_equal(got, want)


@pytest.mark.timeout(150)
@pytest.mark.parametrize("nranks", [1, 2])
def test_dist_coarse_level_v_with_nonzero_boundary_is_zeroed(nranks):
    """The cycle zeroes the coarse v INCLUDING its boundary (setToValue(v, 0, true), N3/MultiGrid3D.cpp:634).  The slab
    driver normally skips the fill (the first red pass does not read v) and relies on the boundary being zero in
    memory; a v uploaded into a coarse distributed level breaks that premise and must bring the fill back."""
    n3 = [65, 65, 65]
    rng = np.random.default_rng(7)
    f0 = rng.uniform(-1, 1, O.shape(n3)).astype(np.float64)
    junk = rng.uniform(-1, 1, O.shape([33] * 3)).astype(np.float64)  # boundary entries included
    got, info = run_ranks(nranks, n3, R3, np.float64, 2, 2, 2, 4, f0=f0, v_levels={1: junk})
    assert info[0][0] >= 2, "level 1 must be a distributed level for this test"
    want = O.cycle3d(n3, R3, mode=0, v1=2, v2=2, reps=2, f=f0, dtype=np.float64)
    assert bits_equal(got, want)


@pytest.mark.timeout(150)
@pytest.mark.parametrize("nranks", [2, 4])
def test_dist_vcycle_random_inputs_f32_and_correct_mode(nranks):
    n3 = [65, 33, 65]  # anisotropic in y: slabs are along z
    rng = np.random.default_rng(nranks)
    rg = [-1, 1, 0, 2, 0.5, 3]
    v0 = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)
    f0 = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)
    for mode in (P.REF_COMPAT, P.CORRECT):
        got, _ = run_ranks(nranks, n3, rg, np.float32, 1, 2, 1, 4, mode=mode, v0=v0, f0=f0)
        want = O.cycle3d(n3, rg, mode=0, v1=1, v2=2, reps=1, v=v0, f=f0, residual_mode=mode, dtype=np.float32)
        assert bits_equal(got, want)


@pytest.mark.timeout(200)
@pytest.mark.parametrize("nranks,dtype", [(2, np.float64), (4, np.float64), (2, np.float32)])
def test_dist_vcycle_wide_rows_pipelined_smoother_on_slabs(nranks, dtype):
    """257 x 129 rows: the slabs' interior planes run the pipelined LDS-exchange smoother (automatic choice), their
    edge planes the small-workgroup kernel; random data so that every ghost plane matters"""
    n3 = [257, 129, 65]
    rng = np.random.default_rng(7 + nranks)
    rg = [-1, 1, 0, 2, 0.5, 3]
    v0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    got, info = run_ranks(nranks, n3, rg, dtype, 2, 2, 1, 8, v0=v0, f0=f0)
    assert info[0][0] >= 1
    want = O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=1, v=v0, f=f0, dtype=dtype)
    assert bits_equal(got, want)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nranks,inline_bytes,v2", [(1, 0, 2), (2, 0, 1), (2, None, 3), (4, 0, 2), (8, None, 2)])
def test_dist_correction_read_on_the_fly_on_slabs(nranks, inline_bytes, v2):
    """levels with >= 257-point rows: the post-smoothing's first red pass reads v + Interpolate(coarse) on the fly on slabs
    too (set P corrected in place incl. the ghost planes, no stored correction, no exchange of corrected planes); edge
    planes, interior and ghost planes must all see the same corrections.  Two cycles, both exchange modes, the coarse
    level distributed (2 ranks) or replicated (4, 8 ranks)."""
    n3 = [257, 129, 129]
    rng = np.random.default_rng(100 + nranks)
    rg = [-1, 1, 0, 2, 0.5, 3]
    v0 = rng.uniform(-1, 1, O.shape(n3))
    f0 = rng.uniform(-1, 1, O.shape(n3))
    names = {}
    got, info = run_ranks(nranks, n3, rg, np.float64, 2, v2, 2, 16, v0=v0, f0=f0, inline_bytes=inline_bytes,
                          extra=lambda mg: names.setdefault(mg.rank, mg.ctx.last_relax_kernel()))
    want = O.cycle3d(n3, rg, mode=0, v1=2, v2=v2, reps=2, v=v0, f=f0, dtype=np.float64)
    assert bits_equal(got, want)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nranks,n3,min_planes,params", [(2, [257, 257, 17], 8, None), (2, [257, 257, 17], 8, {"rr3d.black": 2}),
                                                         (4, [257, 129, 33], 8, None), (8, [257, 129, 65], 8, {"rr3d.black": 2})])
def test_dist_every_rank_takes_the_same_form_of_an_operator(nranks, n3, min_planes, params):
    """slabs of exactly 8 planes: the last rank updates one plane less than the others (its top plane is the boundary), which is below
    the minimum of the correcting pass -- the choice between the forms of an operator (which differ in what they exchange) must not
    depend on the rank's own plane count (found by tests/checkers/fuzz_dist.py: rank 0 waited for a ghost plane rank 1 never sent)"""
    rng = np.random.default_rng(nranks + n3[2])
    rg = [0, 1, 0, 1, 0, 1]
    v0 = rng.uniform(-1, 1, O.shape(n3))
    f0 = rng.uniform(-1, 1, O.shape(n3))
    for inline_bytes in (0, None):
        got, info = run_ranks(nranks, n3, rg, np.float64, 2, 2, 2, min_planes, mode=P.CORRECT, v0=v0, f0=f0, inline_bytes=inline_bytes, params=params)
        want = O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=2, v=v0, f=f0, residual_mode=P.CORRECT, dtype=np.float64)
        assert bits_equal(got, want)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nranks,v1,dtype", [(1, 2, np.float64), (2, 1, np.float64), (2, 2, np.float32), (4, 2, np.float64), (8, 1, np.float64),
                                             (8, 3, np.float32)])
def test_dist_last_black_pass_inside_residual_restrict_on_slabs(nranks, v1, dtype):
    """the way down of a distributed level with the last black pass inside the residual+restrict launch
    (mgx3dxs_relax_rr_slab on the coarse planes whose inputs are local, edge planes by the colour-pass kernel before the
    exchanges, the one or two remaining coarse planes after them): "rr3d.black" = 2 makes these small levels take it.  Two
    cycles from random data (the second one starts from non-trivial v on every level), coarse level distributed (1, 2, 4
    ranks: from-zero pre-smoothing on it) or replicated (8 ranks: the all-gathered share), against the oracle."""
    n3 = [129, 65, 129]
    rng = np.random.default_rng(300 + nranks + v1)
    rg = [-1, 1, 0, 2, 0.5, 3]
    v0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    names = {}
    # ca_min_planes = 0: this test is about the exchange-per-colour-pass schedule (slab_black_rr_); the communication-avoiding
    # schedule has its own suite (tests/test_gpu_dist_ca.py)
    got, info = run_ranks(nranks, n3, rg, dtype, v1, 2, 2, 16, v0=v0, f0=f0, params={"rr3d.black": 2}, ca_min_planes=0,
                          extra=lambda mg: names.setdefault(mg.rank, mg.ctx.last_rr_kernel()))
    assert all(names[r].startswith("relax_rr3d_xs_kernel") for r in range(nranks)), names
    want = O.cycle3d(n3, rg, mode=0, v1=v1, v2=2, reps=2, v=v0, f=f0, dtype=dtype)
    assert bits_equal(got, want)


@pytest.mark.timeout(200)
@pytest.mark.parametrize("nranks", [2, 4])
@pytest.mark.parametrize("inline_bytes", [None, 600_000, 100_000])
def test_dist_inline_and_mixed_exchange_modes(nranks, inline_bytes):
    """levels whose slab is small exchange INLINE on the compute stream (one launch per pass, no cross-stream events):
    the library default (every level of these grids), a threshold between level 0 and level 1 (65^3 / 2 ranks: 1.08 MB
    and 139 KB -- the mode switches in the middle of every cycle, in both directions) and one below most levels; V-cycles
    with random inputs in both residual modes, and FMG -- bit-identical to the single-GPU result"""
    n3 = [65, 33, 65]
    rng = np.random.default_rng(nranks)
    rg = [-1, 1, 0, 2, 0.5, 3]
    for dtype, mode in ((np.float64, P.REF_COMPAT), (np.float32, P.CORRECT)):
        v0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
        f0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
        got, info = run_ranks(nranks, n3, rg, dtype, 2, 1, 2, 2, mode=mode, v0=v0, f0=f0, inline_bytes=inline_bytes)
        want = O.cycle3d(n3, rg, mode=0, v1=2, v2=1, reps=2, v=v0, f=f0, residual_mode=mode, dtype=dtype)
        assert not np.isnan(got).any()
        assert bits_equal(got, want), (dtype, mode)
    got, info = run_ranks(nranks, [65] * 3, R3, np.float64, 1, 2, 0, 4, fmg=2, inline_bytes=inline_bytes)
    assert bits_equal(got, O.cycle3d([65] * 3, R3, mode=1, v0=2, v1=1, v2=2, dtype=np.float64))


@pytest.mark.timeout(200)
@pytest.mark.parametrize("nranks", [1, 2, 4])
@pytest.mark.parametrize("n,min_planes", [(33, 2), (65, 4), (65, 8)])
def test_dist_fmg_matches_oracle(nranks, n, min_planes):
    """FullMultiGridVCycle on slabs (Restrict of f with the ghost below, all-gather into the replicated tail incl. its top
    boundary plane, plain Interpolate on the way up) == oracle, reference and corrected residual"""
    for mode in (P.REF_COMPAT, P.CORRECT):
        got, info = run_ranks(nranks, [n] * 3, R3, np.float64, 2, 2, 0, min_planes, mode=mode, fmg=1)
        want = O.cycle3d([n] * 3, R3, mode=1, v0=1, v1=2, v2=2, residual_mode=mode, dtype=np.float64)
        assert not np.isnan(got).any()
        assert bits_equal(got, want), (mode, info)


@pytest.mark.timeout(200)
def test_dist_fmg_random_rhs_f32_then_vcycle():
    n3 = [65, 33, 65]
    rng = np.random.default_rng(11)
    rg = [-1, 1, 0, 2, 0.5, 3]
    f0 = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)   # non-zero on the boundary: exercises the injected planes
    got, _ = run_ranks(4, n3, rg, np.float32, 1, 2, 0, 4, f0=f0, fmg=2)
    want = O.cycle3d(n3, rg, mode=1, v0=2, v1=1, v2=2, f=f0, dtype=np.float32)
    assert bits_equal(got, want)


@pytest.mark.timeout(200)
def test_dist_vcycle_8_ranks_129(known_answers):
    """8 slabs of 16 planes at 129^3 (the 8-GPU shape of the node), reference semantics in fp32"""
    got, info = run_ranks(8, [129] * 3, R3, np.float32, 2, 2, 1, 4)
    assert info[0][0] == 3  # 129 (16 planes/rank), 65 (8) and 33 (4) stay distributed; 17 and coarser are replicated
    want = O.cycle3d([129] * 3, R3, mode=0, v1=2, v2=2, reps=1, dtype=np.float32)
    assert bits_equal(got, want)


@pytest.mark.timeout(150)
def test_dist_numgrids_override_and_relax_only():
    got, _ = run_ranks(2, [65] * 3, R3, np.float64, 3, 0, 1, 4, nlevels=1)   # numGrids := 1 -> smoother only
    want = O.cycle3d([65] * 3, R3, nlevels=1, mode=0, v1=3, v2=0, dtype=np.float64)
    assert bits_equal(got, want)
    got, _ = run_ranks(2, [65] * 3, R3, np.float64, 2, 2, 1, 4, nlevels=3)   # cycle ends on a distributed level
    want = O.cycle3d([65] * 3, R3, nlevels=3, mode=0, v1=2, v2=2, dtype=np.float64)
    assert bits_equal(got, want)


@pytest.mark.timeout(150)
def test_rccl_plumbing_single_rank():
    """RCCL itself cannot be run across GPUs on the one-GPU test box; this checks what can be checked there:
    unique id, ncclCommInitRank, grouped send/recv on the comm stream, event ordering, teardown."""
    import ctypes as C
    ctx = P.Context(0)
    ctx.comm_init(P.Context.unique_id(), 0, 1)
    x = np.random.default_rng(0).uniform(-1, 1, 1 << 16)
    src, dst = ctx.to_device(x), ctx.malloc(x.nbytes)
    P.check(P.lib.mgx_comm_selftest(ctx._h, src, dst, C.c_size_t(x.size)))
    assert bits_equal(ctx.to_host(dst, x.shape, x.dtype), x)
    # the same with the collectives enqueued on the compute stream (the mode of small slab levels), and back
    for on in (True, False):
        ctx.comm_set_inline(on)
        y = np.random.default_rng(1 + on).uniform(-1, 1, 1 << 16)
        P.check(P.lib.mgx_memcpy_h2d(ctx._h, src, y.ctypes.data_as(C.c_void_p), C.c_size_t(y.nbytes)))
        P.check(P.lib.mgx_comm_selftest(ctx._h, src, dst, C.c_size_t(y.size)))
        assert bits_equal(ctx.to_host(dst, y.shape, y.dtype), y)
    # with a (1-rank) RCCL communicator attached the slab driver still works
    mg = P.DistMultiGrid3D(ctx, [33] * 3, R3, np.float64, min_planes=2)
    mg.VCycle(0, 2, 2)
    full = np.zeros((33, 33, 33))
    mg.download_v_into(0, full)
    assert bits_equal(full, O.cycle3d([33] * 3, R3, mode=0, dtype=np.float64))
    mg.close()
    ctx.free(src)
    ctx.free(dst)
    ctx.close()


@pytest.mark.timeout(200)
@pytest.mark.parametrize("nranks", [2, 4])
def test_dist_missing_wait_is_detected(nranks):
    """NEGATIVE test of the harness: with mgx_comm_wait turned into a no-op (fault injection of the test transport) the
    compute stream runs ahead of the ghost transfers and the result must differ from the oracle; with the waits in
    place the same configuration is bit-identical.  Without this, a missing wait in the overlap schedule could pass
    every test (the transfers of small grids would usually win the race by luck)."""
    n3 = [65, 33, 65]
    rng = np.random.default_rng(5)
    rg = [-1, 1, 0, 2, 0.5, 3]
    v0 = rng.uniform(-1, 1, O.shape(n3))
    f0 = rng.uniform(-1, 1, O.shape(n3))
    want = O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=1, v=v0, f=f0, dtype=np.float64)
    good, _ = run_ranks(nranks, n3, rg, np.float64, 2, 2, 1, 4, v0=v0, f0=f0, delay_us=1000)
    assert bits_equal(good, want)
    bad, _ = run_ranks(nranks, n3, rg, np.float64, 2, 2, 1, 4, v0=v0, f0=f0, delay_us=1000, drop_waits=True)
    assert not np.isnan(bad).any()
    assert not bits_equal(bad, want), "the dropped waits went unnoticed: the transport is not exercising the event ordering"
    good0, _ = run_ranks(nranks, n3, rg, np.float64, 2, 2, 1, 4, v0=v0, f0=f0, delay_us=0)  # and without the delay
    assert bits_equal(good0, want)


@pytest.mark.timeout(200)
@pytest.mark.parametrize("nranks", [1, 2, 4])
@pytest.mark.parametrize("mode", [P.REF_COMPAT, P.CORRECT])
def test_dist_residual_norm_allreduce(nranks, mode):
    """mgDistMultiGrid3D_ResidualNorm: slab sums of the squared residual (device reduction in a fixed order) + one
    all-reduce of one double; every rank gets the same bits.  ADDITION without a reference: parity
    unpinned, checked against numpy on the oracle's residual of the oracle's v (relative 1e-12)."""
    n3 = [65, 33, 65]
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(9)
    v0 = rng.uniform(-1, 1, O.shape(n3))
    f0 = rng.uniform(-1, 1, O.shape(n3))

    def extra(mg):
        a = mg.ResidualNorm(0)
        mg.ResidualNormRecord(0)
        mg.VCycle(0, 1, 1)
        mg.ResidualNormRecord(0)
        return a, mg.ResidualNorm(0), mg.ResidualNormHistory()

    _, info = run_ranks(nranks, n3, rg, np.float64, 2, 2, 1, 4, mode=mode, v0=v0, f0=f0, extra=extra)
    v1 = O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=1, v=v0, f=f0, residual_mode=mode, dtype=np.float64)
    v2 = O.cycle3d(n3, rg, mode=0, v1=1, v2=1, reps=1, v=v1, f=f0, residual_mode=mode, dtype=np.float64)
    want = [float(np.sqrt(np.sum(O.residual3d(n3, rg, v, f0, mode, dtype=np.float64) ** 2))) for v in (v1, v2)]
    got = [info[("extra", r)] for r in range(nranks)]
    for g in got[1:]:
        assert g == got[0], "ranks disagree on the all-reduced norm"
    a, b, hist = got[0]
    assert abs(a - want[0]) <= 1e-12 * want[0] and abs(b - want[1]) <= 1e-12 * want[1]
    assert hist == [a, b]


@pytest.mark.timeout(100)
def test_local_transport_allreduce_and_allgather_bits():
    """the test transport's collectives: all-reduce = sum in rank order on every rank (same bits), all-gather in rank order"""
    import ctypes as C
    nranks, count = 4, 1000
    ctxs = [P.Context(0) for _ in range(nranks)]
    group = P.LocalGroup(nranks)
    group.set_test_hooks(DELAY_US, False)
    for r, c in enumerate(ctxs):
        group.attach(c, r)
    rng = np.random.default_rng(1)
    data = [rng.uniform(-1, 1, count) * 10.0 ** rng.integers(-8, 8, count) for _ in range(nranks)]
    out, errors = {}, []

    def worker(r):
        try:
            c = ctxs[r]
            x = c.to_device(data[r])
            g = c.malloc(nranks * count * 8)
            P.check(P.lib.mgx_comm_allgather(c._h, x, g, C.c_size_t(count), C.c_int(8)))
            P.check(P.lib.mgx_comm_wait(c._h))
            gathered = c.to_host(g, (nranks, count), np.float64)
            P.check(P.lib.mgx_comm_allreduce_sum_f64(c._h, x, C.c_size_t(count)))
            P.check(P.lib.mgx_comm_wait(c._h))
            out[r] = (gathered, c.to_host(x, (count,), np.float64))
            c.free(x)
            c.free(g)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(nranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=60)
    assert not errors, errors
    assert not [t for t in threads if t.is_alive()]
    want = data[0].copy()
    for r in range(1, nranks):
        want = want + data[r]
    for r in range(nranks):
        assert bits_equal(out[r][0], np.stack(data))
        assert bits_equal(out[r][1], want)
    for c in ctxs:
        c.close()
    group.close()


@pytest.mark.timeout(200)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_dist_vcycle_graph_replay_one_rank(dtype):
    """use_graph on the slab hierarchy (opt-in): the cycle from level 0 is captured once per set of host-side flags and replayed;
    five cycles of one rank against the oracle (the first ones capture again while the flags settle)"""
    n3 = [129, 65, 65]
    rg = [-1, 1, 0, 2, 0.5, 3]
    c = P.Context(0)
    try:
        r = np.random.default_rng(8)
        v0 = r.uniform(-1, 1, O.shape(n3)).astype(dtype)
        f0 = r.uniform(-1, 1, O.shape(n3)).astype(dtype)
        mg = P.DistMultiGrid3D(c, n3, rg, dtype, min_planes=8, use_graph=True)
        mg.upload_f(0, f0)
        mg.upload_v(0, v0)
        for _ in range(5):
            mg.VCycle(0, 2, 2)
        got = np.full(O.shape(n3), np.nan, dtype)
        mg.download_v_into(0, got)
        mg.close()
        assert bits_equal(got, O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=5, v=v0, f=f0, dtype=dtype))
    finally:
        c.close()


@pytest.mark.timeout(200)
def test_rehearsed_rank_with_graph_replay_runs():
    """one of eight ranks rehearsed (RCCL self send / recv of every message) with the cycle captured into a HIP graph: the RCCL
    calls are part of the capture.  Values are meaningless in a rehearsal; this checks that capture and replay work at all."""
    c = P.Context(0)
    try:
        c.comm_init_rehearsal(P.Context.unique_id(), 4, 8)
        mg = P.DistMultiGrid3D(c, [129, 129, 257], R3, np.float64, min_planes=8, use_graph=True)
        for _ in range(4):
            mg.VCycle(0, 2, 2)
        c.sync()
        mg.close()
    finally:
        c.close()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nranks,inline_bytes,dtype", [(2, 0, np.float64), (4, None, np.float32), (8, 0, np.float64)])
def test_dist_half_plane_exchanges(nranks, inline_bytes, dtype):
    """pack_halos: behind a colour pass only the half-rows of the colour it changed travel (packed on the compute stream,
    unpacked on the stream of the receive); whole planes where both colours changed.  V(2,2) x 2 and the fused way down
    forced, both exchange modes, against the oracle"""
    n3 = [129, 65, 129]
    rng = np.random.default_rng(500 + nranks)
    rg = [-1, 1, 0, 2, 0.5, 3]
    v0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f0 = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    got, info = run_ranks(nranks, n3, rg, dtype, 2, 2, 2, 8, v0=v0, f0=f0, inline_bytes=inline_bytes, params={"rr3d.black": 2},
                          pack_halos=True)
    assert bits_equal(got, O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=2, v=v0, f=f0, dtype=dtype))
