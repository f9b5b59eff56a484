"""GPU suite: the first sweep of a coarse level's pre-smoothing in ONE launch (relax3d_xs_pipe_kernel, VAR = 3: the level counts
as all zeros, so the red pass's results are relax3d_point(0, ..., 0, f) and the black pass forms them itself from the f it
loads) against the oracle's MultiGrid3D::Relax on a zeroed level (N3/MultiGrid3D.cpp:634 + :489-567), bit for bit: every
word, red entries included."""
import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal

pytestmark = pytest.mark.gpu
RG = [-1, 1, 0, 2, 0.5, 3]


@pytest.fixture(scope="module")
def ctx():
    c = P.Context(0)
    yield c
    c.close()


def _case(ctx, n3, dtype, ncycles, seed, expect_fused=True):
    r = np.random.default_rng(seed)
    shape = tuple(reversed(n3))
    v = r.uniform(-1, 1, shape).astype(dtype)  # junk inside: the level counts as zero; the boundary is zero by contract
    f = r.uniform(-1, 1, shape).astype(dtype)
    v[0] = v[-1] = 0
    v[:, 0] = v[:, -1] = 0
    v[:, :, 0] = v[:, :, -1] = 0
    got = P.ops3dxs.relax_from_zero(ctx, v, f, n3, RG, ncycles, True)
    if ncycles == 1:
        assert ctx.last_relax_kernel().endswith(",3>") == expect_fused, ctx.last_relax_kernel()
    assert bits_equal(got, O.relax3d(n3, RG, np.zeros_like(v), f, ncycles, dtype=dtype))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(257, 129, 17), (257, 257, 33), (513, 129, 17), (257, 129, 129), (1025, 129, 17)])
@pytest.mark.parametrize("ncycles", [1, 2])
def test_first_sweep_from_zero_in_one_launch(ctx, n3, dtype, ncycles):
    # fp32 rows of 513 points and more run the two-pairs-per-lane smoother: not taken there
    _case(ctx, n3, dtype, ncycles, seed=n3[0] + n3[2], expect_fused=not (dtype == np.float32 and n3[0] >= 513))


def test_first_sweep_off_switch_and_run_lengths(ctx):
    for zc in (8, 9, 31):
        ctx.set_param("relax3d.zchunk", zc)
        try:
            _case(ctx, (257, 129, 65), np.float64, 1, seed=zc)
        finally:
            ctx.set_param("relax3d.zchunk", 0)
    ctx.set_param("relax3d.zero_sweep", 0)
    try:
        _case(ctx, (257, 129, 65), np.float64, 1, seed=1, expect_fused=False)
    finally:
        ctx.set_param("relax3d.zero_sweep", 1)


def test_first_sweep_inside_the_way_down(ctx):
    """smooth_residual_restrict from zero with two sweeps on a level that also fuses its last black pass: first sweep in one
    launch, one red pass, black pass + residual + restrict in one launch"""
    n3 = (513, 129, 65)
    r = np.random.default_rng(4)
    shape = tuple(reversed(n3))
    v = np.zeros(shape)
    f = r.uniform(-1, 1, shape)
    got_v, got_c = P.ops3dxs.smooth_residual_restrict(ctx, v, f, n3, RG, 2, True, True, P.REF_COMPAT)
    assert ctx.last_rr_kernel().startswith("relax_rr3d_xs_kernel")
    want_v = O.relax3d(n3, RG, v, f, 2, dtype=np.float64)
    assert bits_equal(got_v, want_v)
    assert bits_equal(got_c, O.restrict3d(n3, O.residual3d(n3, RG, want_v, f, P.REF_COMPAT, dtype=np.float64), dtype=np.float64))
