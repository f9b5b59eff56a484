"""GPU suite: the one-launch red+black sweep (mgx3dxs_relax_pp, csrc/mgx_sweep3d.hip) against the oracle's
MultiGrid3D::Relax restatement (N3/MultiGrid3D.cpp:489-567), bit for bit.  The kernel orders workgroups by progress
words in memory: every case checks EVERY word of the result, with run lengths, leads and sweep counts that move the
hand-offs around (short runs: pipeline fill / drain and the recomputed red planes at run ends dominate)."""
import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal

pytestmark = pytest.mark.gpu
RG = [-1, 1, 0, 2, 0.5, 3]  # anisotropic box: hx != hy != hz


@pytest.fixture(scope="module")
def ctx():
    c = P.Context(0)
    c.set_param("relax3d.fused", 1)
    yield c
    c.close()


def _data(n3, dtype, seed=0):
    r = np.random.default_rng(seed)
    shape = tuple(reversed(n3))
    return r.uniform(-1, 1, shape).astype(dtype), r.uniform(-1, 1, shape).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(513, 129, 129), (513, 257, 129), (513, 129, 513)])
@pytest.mark.parametrize("ncycles", [2, 3])
def test_sweep_matches_oracle(ctx, n3, ncycles, dtype):
    assert P.ops3dxs.relax_pp_takes(ctx, n3, ncycles, dtype)
    v, f = _data(n3, dtype)  # random boundary values too: the partner array's boundary has to be brought along
    got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, ncycles)
    assert ctx.last_relax_kernel().startswith("sweep3d_xs_kernel"), ctx.last_relax_kernel()
    ctx.sync()  # raises if a wait gave up
    want = O.relax3d(n3, RG, v, f, ncycles, dtype=dtype)
    assert bits_equal(got, want)


@pytest.mark.parametrize("zchunk", [2, 3, 7, 8, 9, 16, 33, 127])
def test_sweep_run_lengths(ctx, zchunk):
    """runs shorter than, equal to and longer than the lead of the red stage; first / last run ends on a boundary plane"""
    n3 = (513, 129, 129)
    v, f = _data(n3, np.float64, seed=zchunk)
    want = O.relax3d(n3, RG, v, f, 2, dtype=np.float64)
    if (127 + zchunk - 1) // zchunk * 16 > 256:  # more workgroups than CUs: the library must refuse, not hang
        ctx.set_param("relax3d.zchunk", zchunk)
        try:
            with pytest.raises(P.MgxError):
                P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        finally:
            ctx.set_param("relax3d.zchunk", 0)
        return
    ctx.set_param("relax3d.zchunk", zchunk)
    try:
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        ctx.sync()
    finally:
        ctx.set_param("relax3d.zchunk", 0)
    assert bits_equal(got, want)


@pytest.mark.parametrize("lead", [5, 6, 7])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sweep_leads(ctx, lead, dtype):
    n3 = (513, 257, 129)
    v, f = _data(n3, dtype, seed=lead)
    ctx.set_param("relax3d.fused_lead", lead)
    try:
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 4)
        ctx.sync()
    finally:
        ctx.set_param("relax3d.fused_lead", 0)
    assert bits_equal(got, O.relax3d(n3, RG, v, f, 4, dtype=dtype))


def test_sweep_off_switch_and_small_levels(ctx):
    n3 = (513, 129, 129)
    v, f = _data(n3, np.float64)
    ctx.set_param("relax3d.fused", 0)
    try:
        assert not P.ops3dxs.relax_pp_takes(ctx, n3, 2)
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        assert not ctx.last_relax_kernel().startswith("sweep3d")
    finally:
        ctx.set_param("relax3d.fused", 1)
    assert bits_equal(got, O.relax3d(n3, RG, v, f, 2, dtype=np.float64))
    n3 = (129, 65, 33)  # not a level of 513-point rows: colour passes
    assert not P.ops3dxs.relax_pp_takes(ctx, n3, 2)
    v, f = _data(n3, np.float64)
    assert bits_equal(P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2), O.relax3d(n3, RG, v, f, 2, dtype=np.float64))


def test_sweep_many_launches_epochs(ctx):
    """20 back-to-back launches on one context: the progress words of one launch must never satisfy the next"""
    n3 = (513, 129, 129)
    v, f = _data(n3, np.float64, seed=5)
    got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 20)
    ctx.sync()
    assert bits_equal(got, O.relax3d(n3, RG, v, f, 20, dtype=np.float64))


def test_hierarchy_relax_uses_sweep_and_keeps_boundary(ctx):
    n = 513
    mg = P.MultiGrid3D(ctx, (n, 129, 129), RG, np.float64)
    v, f = _data((n, 129, 129), np.float64, seed=9)
    mg.upload_v(0, v)
    mg.upload_f(0, f)
    mg.Relax(0, 2)
    assert ctx.last_relax_kernel().startswith("sweep3d_xs_kernel")
    mg.Relax(0, 3)  # partner boundary now vouched for; odd count: one sweep as colour passes
    got = mg.download_v(0)
    mg.close()
    assert bits_equal(got, O.relax3d((n, 129, 129), RG, v, f, 5, dtype=np.float64))
