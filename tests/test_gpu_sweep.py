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
    c.set_param("relax3d.resident", 0)  # these cases are about the one-launch-per-sweep kernels (tests/test_gpu_resident.py: the other form)
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


@pytest.mark.parametrize("ilv", [0, 1])
@pytest.mark.parametrize("lead", [5, 6, 7])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_sweep_leads_and_instruction_orders(ctx, lead, dtype, ilv):
    """the lead of the red stage, and both orders of a step's instructions (memory instructions first / in groups between the
    rows of the arithmetic)"""
    n3 = (513, 257, 129)
    v, f = _data(n3, dtype, seed=lead)
    ctx.set_param("relax3d.fused_lead", lead)
    ctx.set_param("relax3d.fused_ilv", ilv)
    try:
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 4)
        ctx.sync()
    finally:
        ctx.set_param("relax3d.fused_lead", 0)
        ctx.set_param("relax3d.fused_ilv", 0)
    assert bits_equal(got, O.relax3d(n3, RG, v, f, 4, dtype=dtype))


@pytest.mark.parametrize("zchunk", [3, 9, 33])
def test_sweep_interleaved_order_run_lengths(ctx, zchunk):
    n3 = (513, 129, 129)
    v, f = _data(n3, np.float64, seed=40 + zchunk)
    ctx.set_param("relax3d.fused_ilv", 1)
    ctx.set_param("relax3d.zchunk", zchunk if zchunk >= 8 else 8)
    try:
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 3)
        ctx.sync()
        assert "true" in ctx.last_relax_kernel().split(",")[-1]
    finally:
        ctx.set_param("relax3d.zchunk", 0)
        ctx.set_param("relax3d.fused_ilv", 0)
    assert bits_equal(got, O.relax3d(n3, RG, v, f, 3, dtype=np.float64))


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
    n3 = (129, 65, 33)  # neither a level of 513-point rows nor one of at most 65: colour passes
    assert not P.ops3dxs.relax_pp_takes(ctx, n3, 2)
    v, f = _data(n3, np.float64)
    assert bits_equal(P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2), O.relax3d(n3, RG, v, f, 2, dtype=np.float64))
    n3 = (17, 17, 17)  # one-workgroup kernel
    assert not P.ops3dxs.relax_pp_takes(ctx, n3, 2)


# ------------------------------------------------------------------ cache-resident levels: one launch per sweep, halo recomputed
MID = [(65, 65, 65), (33, 33, 33), (65, 33, 129), (65, 129, 9), (33, 9, 129), (33, 65, 17)]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", MID)
def test_mid_sweep_matches_oracle(ctx, n3, dtype):
    assert P.ops3dxs.relax_pp_takes(ctx, n3, 2, dtype)
    v, f = _data(n3, dtype, seed=sum(n3))  # random boundary values too
    for ncycles in (2, 3, 4):
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, ncycles)
        assert ctx.last_relax_kernel().startswith("sweep3d_xs_mid_kernel"), ctx.last_relax_kernel()
        assert bits_equal(got, O.relax3d(n3, RG, v, f, ncycles, dtype=dtype)), ncycles


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(65, 65, 65), (65, 33, 129), (33, 33, 33)])
def test_mid_relax_from_zero(ctx, n3, dtype):
    """v counts as zeros: junk in the interior must not matter, the boundary is zero by contract; 2 sweeps: the first does not
    read v; 3 sweeps and a non-zero boundary: the plain path"""
    v, f = _data(n3, dtype, seed=1)
    v[0], v[-1], v[:, 0], v[:, -1], v[:, :, 0], v[:, :, -1] = 0, 0, 0, 0, 0, 0
    zero = np.zeros_like(v)
    for ncycles in (2, 4, 3):
        got = P.ops3dxs.relax_from_zero_pp(ctx, v, f, n3, RG, ncycles, True)
        assert bits_equal(got, O.relax3d(n3, RG, zero, f, ncycles, dtype=dtype)), ncycles
    v2, _ = _data(n3, dtype, seed=2)  # non-zero boundary: rim_is_zero = False -> everything is zeroed first
    got = P.ops3dxs.relax_from_zero_pp(ctx, v2, f, n3, RG, 2, False)
    assert bits_equal(got, O.relax3d(n3, RG, zero, f, 2, dtype=dtype))


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(65, 65, 65), (65, 129, 33), (33, 33, 33), (33, 9, 65)])
def test_mid_single_sweeps_out_of_place(ctx, n3, dtype):
    """the unit the ping-pong drivers are made of, alone and repeatedly on one context (LDS and registers carry the previous
    launch's values): vin -> vout, generic and from-zero form; vout's boundary (NaN here) is neither read nor written"""
    v, f = _data(n3, dtype, seed=11)
    zero = np.zeros_like(v)
    nanw = np.full_like(v, np.nan)
    inner = (slice(1, -1),) * 3
    want, wantz = O.relax3d(n3, RG, v, f, 1, dtype=dtype), O.relax3d(n3, RG, zero, f, 1, dtype=dtype)
    for _ in range(3):
        got = P.ops3dxs.sweep_once(ctx, v, nanw, f, n3, RG)
        assert bits_equal(got[inner], want[inner]) and np.isnan(got[0]).all() and np.isnan(got[:, :, -1]).all()
        got = P.ops3dxs.sweep_once(ctx, v, nanw, f, n3, RG, zero=True)  # v (junk, non-zero boundary) must not matter
        assert bits_equal(got[inner], wantz[inner])


@pytest.mark.parametrize("n3", [(65, 65, 65), (65, 129, 33)])
def test_mid_interpolate_correct_relax(ctx, n3):
    v, f = _data(n3, np.float64, seed=4)
    cn = P.coarse_size(n3)
    c = np.random.default_rng(5).uniform(-1, 1, tuple(reversed(cn)))
    want = O.relax3d(n3, RG, O.correct3d(n3, v, O.interpolate3d(n3, v, c, dtype=np.float64), dtype=np.float64), f, 2, dtype=np.float64)
    assert bits_equal(P.ops3dxs.interpolate_correct_relax_pp(ctx, v, f, n3, RG, c, 2), want)


def test_mid_off_switch(ctx):
    n3 = (65, 65, 33)
    v, f = _data(n3, np.float64)
    ctx.set_param("relax3d.fused_mid", 0)
    try:
        assert not P.ops3dxs.relax_pp_takes(ctx, n3, 2)
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        assert not ctx.last_relax_kernel().startswith("sweep3d")
    finally:
        ctx.set_param("relax3d.fused_mid", 1)
    assert bits_equal(got, O.relax3d(n3, RG, v, f, 2, dtype=np.float64))


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


def test_lds_poisoning_reaches_every_cu(ctx):
    """the suite runs with mgx_test_set_lds_poison(1) (tests/conftest.py): every launch is preceded by a launch that fills each
    CU's LDS with NaN patterns.  Self-test: a probe that reads all 160 KB of LDS on every CU without writing sees nothing but the
    pattern -- so a kernel that reads an LDS word before writing it computes with NaNs and fails its parity test."""
    import ctypes
    import os
    if os.environ.get("MGX_POISON_LDS", "1") == "0":
        pytest.skip("poisoning switched off")
    frac = ctypes.c_double(-1.0)
    for _ in range(3):
        P.check(P.lib.mgx_test_lds_probe(ctx._h, ctypes.byref(frac)))
        assert frac.value == 1.0, frac.value
    P.lib.mgx_test_set_lds_poison(0)
    try:
        ctx.sync()
        P.check(P.lib.mgx_test_lds_probe(ctx._h, ctypes.byref(frac)))  # the previous probe wrote nothing: still all pattern
        assert frac.value == 1.0, frac.value
        v, f = _data((65, 65, 65), np.float64, seed=3)
        P.ops3dxs.relax_pp(ctx, v, f, (65, 65, 65), RG, 2)  # a real kernel overwrites LDS on the CUs it runs on
        P.check(P.lib.mgx_test_lds_probe(ctx._h, ctypes.byref(frac)))
        assert frac.value < 1.0, frac.value
    finally:
        P.lib.mgx_test_set_lds_poison(1)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_mid_from_zero_specialisation_at_129_rows(ctx, dtype):
    """the case that failed in round 3 with a template-specialised from-zero form of sweep3d_xs_mid_kernel ("wrong values next to
    boundary faces whenever the previous launch had left non-zero data in LDS"): 129^3, two sweeps from zero.  The kernel is
    specialised on ZERO again (nothing staged, B never read); with LDS poisoned before every launch a read of an unwritten word
    cannot pass.  "relax3d.fused_mid" = 2 lets rows of 129 points into the kernel (the default keeps them on colour passes)."""
    n3 = (129, 129, 129)
    v, f = _data(n3, dtype, seed=1)
    v[0], v[-1], v[:, 0], v[:, -1], v[:, :, 0], v[:, :, -1] = 0, 0, 0, 0, 0, 0
    zero = np.zeros_like(v)
    ctx.set_param("relax3d.fused_mid", 2)
    ctx.set_param("relax3d.resident", 0)
    try:
        for ncycles in (2, 4):
            got = P.ops3dxs.relax_from_zero_pp(ctx, v, f, n3, RG, ncycles, True)
            assert ctx.last_relax_kernel().startswith("sweep3d_xs_mid_kernel"), ctx.last_relax_kernel()
            assert bits_equal(got, O.relax3d(n3, RG, zero, f, ncycles, dtype=dtype)), ncycles
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        assert ctx.last_relax_kernel().startswith("sweep3d_xs_mid_kernel"), ctx.last_relax_kernel()
        assert bits_equal(got, O.relax3d(n3, RG, v, f, 2, dtype=dtype))
    finally:
        ctx.set_param("relax3d.fused_mid", 1)
        ctx.set_param("relax3d.resident", 1)
