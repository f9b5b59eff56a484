"""GPU suite: the pipelined smoother with its step loop unrolled four times ("relax3d.unroll": register roles and row parity
fixed per step, loads and stores through buffer descriptors; csrc/mgx_pipe_step.inc).  Same loads, same stores, same arithmetic
as the rolled loop: every result bit-identical to the oracle's MultiGrid3D::Relax (N3/MultiGrid3D.cpp:489-567) and, for the
correcting red pass, to Relax(ApplyCorrection(Interpolate)) (:638-645).  Runs of every length modulo 4, both entry parities,
odd run lengths asked for (the launch rounds them up to even ones)."""
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


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(257, 129, 33), (513, 129, 17), (257, 257, 21), (1025, 129, 17), (257, 65, 9), (513, 513, 37)])
def test_unrolled_correcting_pass(ctx, dtype, n3):
    rng = np.random.default_rng(sum(n3) + 7)
    cn = P.coarse_size(n3)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
    corrected = O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3), dtype), c, dtype=dtype), dtype=dtype)
    try:
        ctx.set_param("relax3d.unroll", 15)
        ctx.set_param("relax3d.corr_v2", 0)  # the one-pair-per-lane kernel in fp32 too
        for zchunk in (0, 3, 4, 6, 7, 64):
            ctx.set_param("relax3d.zchunk", zchunk)
            for k in (1, 2):
                want = O.relax3d(n3, RG, corrected, f, k, dtype=dtype)
                assert bits_equal(P.ops3dxs.interpolate_correct_relax(ctx, v, f, n3, RG, c, k), want), (zchunk, k)
    finally:
        ctx.set_param("relax3d.unroll", 7)
        ctx.set_param("relax3d.corr_v2", 1)
        ctx.set_param("relax3d.zchunk", 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(257, 129, 33), (513, 65, 13), (257, 257, 7), (1025, 129, 17), (513, 513, 39)])
def test_unrolled_plain_pass(ctx, dtype, n3):
    rng = np.random.default_rng(sum(n3) + 11)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    try:
        ctx.set_param("relax3d.v2", 0)
        ctx.set_param("relax3d.lds", 3282)  # the 2 x 8 x 2 shape with non-temporal f on every size
        for unroll in (15, 31):  # 31: the column and f requested two steps ahead (six steps per loop trip)
            ctx.set_param("relax3d.unroll", unroll)
            for zchunk in (0, 2, 3, 5, 8):
                ctx.set_param("relax3d.zchunk", zchunk)
                for k in (1, 3):
                    want = O.relax3d(n3, RG, v, f, k, dtype=dtype)
                    assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, RG, k), want), (unroll, zchunk, k)
                    assert ctx.last_relax_kernel().startswith("relax3d_xs_pipe_kernel"), ctx.last_relax_kernel()
    finally:
        ctx.set_param("relax3d.unroll", 7)
        ctx.set_param("relax3d.v2", 1)
        ctx.set_param("relax3d.lds", -1)
        ctx.set_param("relax3d.zchunk", 0)


@pytest.mark.parametrize("n3", [(513, 129, 17), (513, 257, 33), (1025, 129, 21), (2049, 129, 9), (513, 65, 39)])
def test_unrolled_two_pair_kernels_fp32(ctx, n3):
    """relax3d_xs_pipe_v2_kernel (fp32, rows of >= 513 points, two x-pairs per lane), plain and correcting pass, unrolled: the two
    points of a row are one packed vector expression there"""
    dtype = np.float32
    rng = np.random.default_rng(sum(n3) + 13)
    cn = P.coarse_size(n3)
    v, f = (rng.uniform(-1, 1, O.shape(n3)).astype(dtype) for _ in range(2))
    c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
    corrected = O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3), dtype), c, dtype=dtype), dtype=dtype)
    try:
        ctx.set_param("relax3d.unroll", 15)
        for zchunk in (0, 2, 3, 5, 8):
            ctx.set_param("relax3d.zchunk", zchunk)
            for k in (1, 2):
                assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, RG, k), O.relax3d(n3, RG, v, f, k, dtype=dtype)), (zchunk, k)
                if n3[2] - 2 >= 8 and n3[1] - 2 >= 64:
                    assert ctx.last_relax_kernel().startswith("relax3d_xs_pipe_v2_kernel"), ctx.last_relax_kernel()
                want = O.relax3d(n3, RG, corrected, f, k, dtype=dtype)
                assert bits_equal(P.ops3dxs.interpolate_correct_relax(ctx, v, f, n3, RG, c, k), want), (zchunk, k)
    finally:
        ctx.set_param("relax3d.unroll", 7)
        ctx.set_param("relax3d.zchunk", 0)


def test_unrolled_tiny_results_take_the_division(ctx):
    """fp32: a quotient below FLT_MIN (zero included) is divided for real -- in the packed form both elements of the row then are"""
    n3 = (513, 65, 9)
    rng = np.random.default_rng(5)
    v = (rng.uniform(-1, 1, O.shape(n3)) * 1e-38).astype(np.float32)
    f = (rng.uniform(-1, 1, O.shape(n3)) * 1e-36).astype(np.float32)
    v[:, :, 100:200] = 0
    f[:, :, 100:200] = 0
    try:
        for u in (15, 0):
            ctx.set_param("relax3d.unroll", u)
            assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, RG, 2), O.relax3d(n3, RG, v, f, 2, dtype=np.float32)), u
    finally:
        ctx.set_param("relax3d.unroll", 7)
