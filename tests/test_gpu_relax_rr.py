"""GPU suite: the way down on one level in one call (mgx3dxs_smooth_residual_restrict, csrc/mgx_relax_rr3d.hip) -- the LAST
BLACK PASS of Relax inside the CalculateResidual + Restrict launch -- against the oracle's restatements of
MultiGrid3D::Relax, ::CalculateResidual and ::Restrict (N3/MultiGrid3D.cpp:489-567, :678-730, :50-184) applied one after
the other, bit for bit: every word of the smoothed v AND of the restricted residual.  "rr3d.black" = 2 makes the kernel take
every geometry, so the cases put grid faces, tile rims (61 coarse columns, 14 / 10 coarse rows per workgroup), halo waves
and the ends of the runs of planes everywhere."""
import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal

pytestmark = pytest.mark.gpu
RG = [-1, 1, 0, 2, 0.5, 3]  # anisotropic box, spacings that are no powers of two: the residual divides
R3 = [0, 1, 0, 1, 0, 1]     # unit cube on 2^k + 1 points: the residual multiplies by exact reciprocals


@pytest.fixture(scope="module")
def ctx():
    c = P.Context(0)
    c.set_param("rr3d.black", 2)
    yield c
    c.close()


def _data(n3, dtype, seed=0):
    r = np.random.default_rng(seed)
    shape = tuple(reversed(n3))
    return r.uniform(-1, 1, shape).astype(dtype), r.uniform(-1, 1, shape).astype(dtype)


def _want(n3, rg, v, f, ncycles, mode, dtype, from_zero=False):
    if from_zero:
        v = np.zeros_like(v)
    w = O.relax3d(n3, rg, v, f, ncycles, dtype=dtype)
    return w, O.restrict3d(n3, O.residual3d(n3, rg, w, f, mode, dtype=dtype), dtype=dtype)


def _check(ctx, n3, rg, ncycles, mode, dtype, seed=0, from_zero=False, v_rim_is_zero=False, fused=True):
    v, f = _data(n3, dtype, seed)
    if v_rim_is_zero:
        v[0] = v[-1] = 0
        v[:, 0] = v[:, -1] = 0
        v[:, :, 0] = v[:, :, -1] = 0
    got_v, got_c = P.ops3dxs.smooth_residual_restrict(ctx, v, f, n3, rg, ncycles, from_zero, v_rim_is_zero, mode)
    assert ctx.last_rr_kernel().startswith("relax_rr3d_xs_kernel") == fused, ctx.last_rr_kernel()
    want_v, want_c = _want(n3, rg, v, f, ncycles, mode, dtype, from_zero)
    assert bits_equal(got_v, want_v)
    assert bits_equal(got_c, want_c)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(9, 9, 9), (17, 9, 33), (33, 65, 17), (65, 33, 33), (129, 129, 65), (257, 65, 33), (513, 33, 17), (127 * 2 - 1, 59, 21)])
def test_every_geometry_random_boundary_values(ctx, n3, dtype):
    """x tiles of 61 coarse columns (1 .. 5 of them), y tiles of 14 coarse rows, faces everywhere; random Dirichlet values"""
    _check(ctx, n3, RG, 1, P.REF_COMPAT, dtype, seed=n3[0])


@pytest.mark.parametrize("mode", [P.REF_COMPAT, P.CORRECT])
@pytest.mark.parametrize("rg", [RG, R3])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_residual_modes_and_reciprocal_spacings(ctx, mode, rg, dtype):
    _check(ctx, (129, 65, 65), rg, 2, mode, dtype, seed=7)


@pytest.mark.parametrize("waves", [12, 16])
@pytest.mark.parametrize("pzchunk", [1, 2, 3, 5, 8, 31])
def test_runs_of_planes_and_workgroup_shapes(ctx, pzchunk, waves):
    """runs of 1 .. all coarse planes: the planes a run relaxes before its first residual, the joints between runs"""
    ctx.set_param("residual_restrict3d.pzchunk", pzchunk)
    ctx.set_param("rr3d.black_waves", waves)
    try:
        _check(ctx, (129, 65, 65), R3, 1, P.REF_COMPAT, np.float64, seed=pzchunk)
        assert ctx.last_rr_kernel().endswith(",%d>" % waves)
        _check(ctx, (65, 129, 33), RG, 1, P.CORRECT, np.float32, seed=pzchunk + 100)
    finally:
        ctx.set_param("residual_restrict3d.pzchunk", 0)
        ctx.set_param("rr3d.black_waves", 0)


@pytest.mark.parametrize("ncycles", [1, 2, 3])
@pytest.mark.parametrize("v_rim_is_zero", [False, True])
def test_from_zero(ctx, ncycles, v_rim_is_zero):
    """the pre-smoothing of a coarse level: v counts as zero (given garbage inside), with and without a zero boundary in memory"""
    _check(ctx, (129, 129, 65), R3, ncycles, P.REF_COMPAT, np.float64, seed=ncycles, from_zero=True, v_rim_is_zero=v_rim_is_zero)
    _check(ctx, (65, 65, 65), RG, ncycles, P.REF_COMPAT, np.float32, seed=ncycles, from_zero=True, v_rim_is_zero=v_rim_is_zero)


def test_off_switch_and_small_levels_run_the_operators_one_by_one(ctx):
    ctx.set_param("rr3d.black", 0)
    try:
        _check(ctx, (129, 65, 65), R3, 2, P.REF_COMPAT, np.float64, fused=False)
    finally:
        ctx.set_param("rr3d.black", 1)
    try:
        _check(ctx, (129, 65, 65), R3, 2, P.REF_COMPAT, np.float64, fused=False)  # automatic: below the HBM-bound sizes
        _check(ctx, (513, 129, 65), R3, 1, P.REF_COMPAT, np.float64, fused=True)
        _check(ctx, (513, 129, 65), R3, 1, P.REF_COMPAT, np.float32, fused=False)  # fp32: measured no faster, left alone
    finally:
        ctx.set_param("rr3d.black", 2)
    v, f = _data((33, 33, 33), np.float64)
    got_v, got_c = P.ops3dxs.smooth_residual_restrict(ctx, v, f, (33, 33, 33), RG, 0)  # no sweep: nothing to fuse
    assert ctx.last_rr_kernel() == ""
    assert bits_equal(got_v, v)
    assert bits_equal(got_c, O.restrict3d((33, 33, 33), O.residual3d((33, 33, 33), RG, v, f, P.REF_COMPAT, dtype=np.float64), dtype=np.float64))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_full_rows_of_the_headline_level(ctx, dtype):
    """513-point rows and 257 rows (five x tiles, nineteen y tiles, several runs): the shape of the 513^3 level, fewer planes"""
    _check(ctx, (513, 513, 33), R3, 2, P.REF_COMPAT, dtype, seed=3)
