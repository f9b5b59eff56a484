"""GPU suite: all colour passes of a Relax call in ONE launch on the cache-resident levels (relax3d_xs_resident2_kernel: the
tiles exchange once per sweep and relax the red points of their first halo ring themselves; relax3d_xs_resident_kernel: once per
pass; csrc/mgx_resident3d.hip) against the oracle's MultiGrid3D::Relax restatement (N3/MultiGrid3D.cpp:489-567), bit for bit.
The workgroups hand their face lines to each other through memory, ordered by progress words: every case checks every
word of the result; sweep counts from 1 to a few hundred move the hand-offs and the double-buffered exchange around, tile
counts from 1 x 1 to 16 x 16 put faces, partial tiles and the extra boundary entry of 129-point rows everywhere."""
import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal

pytestmark = pytest.mark.gpu
RG = [-1, 1, 0, 2, 0.5, 3]


FORMS = {1: "relax3d_xs_resident2_kernel", 2: "relax3d_xs_resident_kernel"}  # "relax3d.resident": one exchange per sweep (default) / per pass


@pytest.fixture(scope="module", params=[(1, 0), (1, 8), (2, 0)], ids=["per_sweep", "per_sweep_tiles8", "per_pass"])
def ctx(request):
    c = P.Context(0)
    c.form, tile = request.param
    c.kernel = FORMS[c.form]
    c.set_param("relax3d.resident", c.form)
    c.set_param("relax3d.resident_tile", tile)  # 0: tiles of 4 x 4 lines where the level has at most one per CU then, else 8 x 8
    c.set_param("relax3d.resident_min", 1)
    yield c
    c.close()


def _kernel(ctx, ncycles):
    """a call of one sweep always runs the per-pass form (its one exchange orders the neighbours' halo loads before the write-back)"""
    return ctx.kernel if ncycles >= 2 else FORMS[2]


def _data(n3, dtype, seed=0):
    r = np.random.default_rng(seed)
    shape = tuple(reversed(n3))
    return r.uniform(-1, 1, shape).astype(dtype), r.uniform(-1, 1, shape).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(33, 33, 33), (65, 65, 65), (129, 129, 129), (129, 65, 33), (65, 129, 17), (33, 9, 129), (129, 17, 9)])
@pytest.mark.parametrize("ncycles", [1, 2, 5])
def test_resident_relax_matches_oracle(ctx, n3, ncycles, dtype):
    v, f = _data(n3, dtype, seed=n3[0] + ncycles)  # random boundary values too
    got = P.ops3dxs.relax(ctx, v, f, n3, RG, ncycles)
    assert ctx.last_relax_kernel().startswith(_kernel(ctx, ncycles)), ctx.last_relax_kernel()
    ctx.sync()  # raises if a wait gave up
    assert bits_equal(got, O.relax3d(n3, RG, v, f, ncycles, dtype=dtype))


@pytest.mark.parametrize("n3,ncycles", [((65, 65, 65), 300), ((129, 129, 129), 40), ((33, 33, 33), 1000)])
def test_resident_long_relax_calls(ctx, n3, ncycles):
    """the reference's own workloads call Relax with thousands of sweeps: hundreds of hand-offs per workgroup in one launch"""
    v, f = _data(n3, np.float32, seed=ncycles)
    got = P.ops3dxs.relax(ctx, v, f, n3, [0, 1, 0, 1, 0, 1], ncycles)
    assert ctx.last_relax_kernel().startswith(ctx.kernel)
    ctx.sync()
    assert bits_equal(got, O.relax3d(n3, [0, 1, 0, 1, 0, 1], v, f, ncycles, dtype=np.float32))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("ncycles", [1, 2, 3])
def test_resident_from_zero(ctx, ncycles, dtype):
    n3 = (129, 65, 65)
    v, f = _data(n3, dtype, seed=9)
    v[0] = v[-1] = 0
    v[:, 0] = v[:, -1] = 0
    v[:, :, 0] = v[:, :, -1] = 0
    got = P.ops3dxs.relax_from_zero(ctx, v, f, n3, RG, ncycles, True)
    assert ctx.last_relax_kernel().startswith(_kernel(ctx, ncycles))
    ctx.sync()
    assert bits_equal(got, O.relax3d(n3, RG, np.zeros_like(v), f, ncycles, dtype=dtype))


def test_resident_many_launches_on_one_context_and_switches(ctx):
    """the launch epoch: twenty launches in a row, other kernels in between; the off switch and the sweep-count threshold"""
    n3 = (65, 65, 33)
    v, f = _data(n3, np.float64, seed=1)
    want = v
    for k in range(20):
        v = P.ops3dxs.relax(ctx, v, f, n3, RG, 2)
        want = O.relax3d(n3, RG, want, f, 2, dtype=np.float64)
    ctx.sync()
    assert bits_equal(v, want)
    ctx.set_param("relax3d.resident", 0)
    try:
        got = P.ops3dxs.relax(ctx, v, f, n3, RG, 2)
        assert not ctx.last_relax_kernel().startswith("relax3d_xs_resident")
        assert bits_equal(got, O.relax3d(n3, RG, v, f, 2, dtype=np.float64))
    finally:
        ctx.set_param("relax3d.resident", ctx.form)
    ctx.set_param("relax3d.resident_min", 3)
    try:
        P.ops3dxs.relax(ctx, v, f, n3, RG, 2)
        assert not ctx.last_relax_kernel().startswith("relax3d_xs_resident")
        P.ops3dxs.relax(ctx, v, f, n3, RG, 3)
        assert ctx.last_relax_kernel().startswith(ctx.kernel)
    finally:
        ctx.set_param("relax3d.resident_min", 1)
    big, fb = _data((257, 33, 33), np.float64)
    P.ops3dxs.relax(ctx, big, fb, (257, 33, 33), RG, 4)  # rows too long for one wave: the colour-pass kernels
    assert not ctx.last_relax_kernel().startswith("relax3d_xs_resident")


@pytest.mark.parametrize("dtype,mode", [(np.float32, 0), (np.float64, 0), (np.float32, 1)])
def test_resident_inside_the_hierarchy(dtype, mode):
    """V(3,3) cycles and FMG(1,3,3) on a 65^3 hierarchy with the library's defaults (three sweeps per call: the 65^3 and
    33^3 levels run their Relax calls in the resident kernel, from zero on the way down), against the oracle's cycle"""
    n3 = [65, 65, 65]
    rg = [-1, 1, 0, 2, 0.5, 3]
    c = P.Context(0)
    try:
        r = np.random.default_rng(5)
        v0 = r.uniform(-1, 1, O.shape(n3)).astype(dtype)
        f0 = r.uniform(-1, 1, O.shape(n3)).astype(dtype)
        mg = P.MultiGrid3D(c, n3, rg, dtype)
        mg.upload_v(0, v0)
        mg.upload_f(0, f0)
        if mode:
            mg.FullMultiGridVCycle(0, 1, 3, 3)
        else:
            mg.VCycle(0, 3, 3)
            mg.VCycle(0, 3, 3)
        assert c.last_relax_kernel().startswith("relax3d_xs_resident2_kernel"), c.last_relax_kernel()
        got = mg.download_v(0)
        mg.close()
        c.sync()
        want = O.cycle3d(n3, rg, mode=mode, v0=1, v1=3, v2=3, reps=1 if mode else 2, v=v0, f=f0, dtype=dtype)
        assert bits_equal(got, want)
    finally:
        c.close()


def test_resident_inside_a_captured_cycle_on_a_fresh_context():
    """use_graph with V(3,3) on a context that has run nothing yet: the hierarchy allocates the hand-off buffers when it is
    created, so the first cycle -- resident Relax launches included -- can be captured; replays advance the launch epoch on
    the device"""
    n3 = [65, 65, 65]
    c = P.Context(0)
    try:
        mg = P.MultiGrid3D(c, n3, RG, np.float32)
        mg.use_graph = True
        for _ in range(4):
            mg.VCycle(0, 3, 3)
        assert c.last_relax_kernel().startswith("relax3d_xs_resident2_kernel")
        got = mg.download_v(0)
        mg.close()
        c.sync()
        assert bits_equal(got, O.cycle3d(n3, RG, mode=0, v1=3, v2=3, reps=4, dtype=np.float32))
    finally:
        c.close()


@pytest.mark.parametrize("n,sweeps", [(33, 300), (65, 120)])
def test_thesis_shaped_fmg_with_long_relax_calls(n, sweeps):
    """the shape of the reference's own workload (N3/Poisson3DSolver.cpp:18-20: FMG(2, 3000, 3000) on the unit cube, fp32, analytic
    right-hand side) with fewer sweeps per call so that the oracle finishes in seconds: every level of 33 ... 65 points runs its
    Relax calls in the resident kernel (one exchange per sweep, from zero on the way down), the levels below in the one-workgroup
    kernel; the solution against the oracle's, bit for bit"""
    n3 = [n] * 3
    c = P.Context(0)
    try:
        mg = P.MultiGrid3D(c, n3, [0, 1, 0, 1, 0, 1], np.float32)
        mg.FullMultiGridVCycle(0, 2, sweeps, sweeps)
        got = mg.download_v(0)
        mg.close()
        c.sync()
        want = O.cycle3d(n3, [0, 1, 0, 1, 0, 1], mode=1, v0=2, v1=sweeps, v2=sweeps, dtype=np.float32)
        assert bits_equal(got, want)
    finally:
        c.close()


@pytest.mark.parametrize("n", [65, 129])
def test_published_workload_full_parameters_known_answer(n):
    """the reference's published 3D workload with its own parameters (thesis Fig. 4.4 = BASELINE.md section 1; N3/Poisson3DSolver.cpp:18-20:
    FMG with 2 V-cycles per level, 3000 + 3000 sweeps per visit, fp32, unit cube) against the hash of the oracle's result (generated once
    by oracle/gen_known_f64.py thesis: ten CPU minutes at n = 129; restatement<float> = the compiled reference's bits)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers_f64.json")) as fh:
        ka = json.load(fh)["3d_n%d_fmg_2_3000_3000_f32" % n]
    c = P.Context(0)
    try:
        mg = P.MultiGrid3D(c, [n] * 3, [0, 1, 0, 1, 0, 1], np.float32)
        assert mg.numGrids == ka["nlevels"]
        mg.FullMultiGridVCycle(0, 2, 3000, 3000)
        got = mg.download_v(0)
        mg.close()
        c.sync()
        assert O.fnv(got) == ka["fnv"] and float(got[n // 2, n // 2, n // 2]) == ka["centre"]
    finally:
        c.close()
