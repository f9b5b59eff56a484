"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against
  (1) the committed golden fixtures = outputs of the unmodified compiled reference (fp32, bit-exact),
  (2) the oracle's double instantiation on seeded inputs (fp64; bar 1e-10 relative L2 from
      BASELINE.json's north_star, expected and asserted: bit-exact),
  (3) known-answer hashes of the reference on the BASELINE.json configurations,
  (4) size-independent properties at full size.
Nothing here reads /root/reference."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal, load_golden, rel_l2

pytestmark = pytest.mark.gpu
A2 = [-1.0, -2.0, 0.0, -3.0]
R3 = [0, 1, 0, 1, 0, 1]
TOL64 = 1e-10  # north_star: "within 1e-10 relative L2 on the solution vector"


@pytest.fixture(scope="module")
def ctx():
    c = P.Context(0)
    yield c
    c.close()


def assert_f64(a, b):
    assert rel_l2(a, b) <= TOL64
    assert bits_equal(a, b), "fp64 result within 1e-10 but not bit-identical"


# ------------------------------------------------------------------ 3D per-operator, fp32 golden
OPS3 = {"natural": P.ops3d, "xsplit": P.ops3dxs}  # the reference layout and the device-internal x-split layout


@pytest.mark.parametrize("layout", ["natural", "xsplit"])
@pytest.mark.parametrize("n", [5, 9, 17])
def test_3d_ops_f32_vs_reference_fixtures(ctx, n, layout):
    ops = OPS3[layout]
    g = load_golden("ops3d_n%d.npz" % n)
    n3, rg, v, f, c = g["n"].tolist(), g["range"].tolist(), g["v"], g["f"], g["c"]
    assert bits_equal(ops.relax(ctx, v, f, n3, rg, 1), g["relax1"])
    assert bits_equal(ops.relax(ctx, v, f, n3, rg, 3), g["relax3"])
    assert bits_equal(ops.residual(ctx, v, f, n3, rg), g["residual"])
    assert bits_equal(ops.restrict(ctx, v, n3), g["restrict"])
    assert bits_equal(ops.interpolate(ctx, v, n3, c), g["interpolate"])
    assert bits_equal(ops.apply_correction(ctx, v, n3, f), g["correct"])
    assert bits_equal(ops.set(ctx, v, n3, 2.5, False), g["set_interior"])
    assert bits_equal(ops.set(ctx, v, n3, 2.5, True), g["set_all"])
    # fused forms == the two reference calls they replace
    assert bits_equal(ops.residual_restrict(ctx, v, f, n3, rg), O.restrict3d(n3, g["residual"]))
    assert bits_equal(ops.interpolate_correct(ctx, v, n3, c), O.correct3d(n3, v, O.interpolate3d(n3, v, c)))
    if layout == "xsplit":
        return
    # cycles through the C host layer
    assert bits_equal(P.solve3d(ctx, v, f, rg, ncycles=1), g["vcycle22"])
    assert bits_equal(P.solve3d(ctx, v, f, rg, fmg=True, v0=1), g["fmg122"])


@pytest.mark.parametrize("layout", ["natural", "xsplit"])
@pytest.mark.parametrize("n3", [(9, 9, 9), (33, 17, 9), (65, 33, 129), (129, 129, 5), (257, 9, 17)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_3d_ops_vs_oracle_random(ctx, n3, dtype, layout):
    """anisotropic sizes and ranges: catches axis swaps the symmetric test problem cannot"""
    ops = OPS3[layout]
    rng = np.random.default_rng(sum(n3))
    rg = [-1, 1, 0, 2, 0.5, 3]
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(O.csize(n3))).astype(dtype)
    eq = bits_equal
    for k in (1, 2):
        assert eq(ops.relax(ctx, v, f, n3, rg, k), O.relax3d(n3, rg, v, f, k, dtype=dtype))
    for mode in (P.REF_COMPAT, P.CORRECT):
        r = O.residual3d(n3, rg, v, f, mode, dtype=dtype)
        assert eq(ops.residual(ctx, v, f, n3, rg, mode), r)
        assert eq(ops.residual_restrict(ctx, v, f, n3, rg, mode), O.restrict3d(n3, r, dtype=dtype))
    assert eq(ops.restrict(ctx, v, n3), O.restrict3d(n3, v, dtype=dtype))
    assert eq(ops.interpolate(ctx, v, n3, c), O.interpolate3d(n3, v, c, dtype=dtype))
    assert eq(ops.interpolate_correct(ctx, v, n3, c), O.correct3d(n3, v, O.interpolate3d(n3, v, c, dtype=dtype), dtype=dtype))
    assert eq(ops.apply_correction(ctx, v, n3, f), O.correct3d(n3, v, f, dtype=dtype))
    for b in (0, 1):
        assert eq(ops.set(ctx, v, n3, -7.25, b), O.set3d(n3, v, -7.25, b, dtype=dtype))


@pytest.mark.parametrize("layout", ["natural", "xsplit"])
def test_3d_smallest_grid_and_zero_sweeps(ctx, layout):
    ops = OPS3[layout]
    n3, rg = [3, 3, 3], R3
    rng = np.random.default_rng(3)
    v = rng.uniform(-1, 1, (3, 3, 3))
    f = rng.uniform(-1, 1, (3, 3, 3))
    assert_f64(ops.relax(ctx, v, f, n3, rg, 4), O.relax3d(n3, rg, v, f, 4, dtype=np.float64))
    assert bits_equal(ops.relax(ctx, v, f, n3, rg, 0), v)
    assert_f64(ops.residual(ctx, v, f, n3, rg), O.residual3d(n3, rg, v, f, dtype=np.float64))


@pytest.mark.parametrize("n3", [(9, 9, 9), (33, 17, 65), (129, 65, 17), (257, 33, 9), (513, 9, 17), (513, 129, 17)])
@pytest.mark.parametrize("stream", [0, 1, 2, 3])
def test_3d_xsplit_residual_restrict_variants(ctx, n3, stream):
    """the x-split residual+restrict kernels (LDS rolling window / streaming shuffles / streaming with the shared rows
    handed over through LDS) == oracle, all chunkings and workgroup heights"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(sum(n3) + stream)
    ctx.set_param("residual_restrict3d.stream", stream)
    try:
        for dtype in (np.float32, np.float64):
            v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
            f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
            for mode in (P.REF_COMPAT, P.CORRECT):
                want = O.restrict3d(n3, O.residual3d(n3, rg, v, f, mode, dtype=dtype), dtype=dtype)
                for chunk in (0, 1, 3):
                    for tyw, rows in ((4, 4), (2, 4), (8, 4), (4, 2), (8, 2)):
                        ctx.set_param("residual_restrict3d.pzchunk", chunk)
                        ctx.set_param("residual_restrict3d.tyw", tyw)
                        ctx.set_param("residual_restrict3d.rows", rows)
                        assert bits_equal(P.ops3dxs.residual_restrict(ctx, v, f, n3, rg, mode), want), (dtype, mode, chunk, tyw, rows)
    finally:
        ctx.set_param("residual_restrict3d.stream", 3)
        ctx.set_param("residual_restrict3d.pzchunk", 0)
        ctx.set_param("residual_restrict3d.tyw", 4)
        ctx.set_param("residual_restrict3d.rows", 0)


@pytest.mark.parametrize("n3", [(9, 9, 9), (65, 33, 17), (129, 129, 33), (513, 129, 17), (513, 257, 33), (513, 513, 17)])
def test_3d_xsplit_residual_restrict_power_of_two_spacings(ctx, n3):
    """squared spacings that are powers of two (the unit cube, [-1,1] x [0,2] x [0,4]): the residual multiplies by the exact
    reciprocals instead of dividing (residual_restrict3d.rcp, default on) == dividing == oracle; a box with one spacing
    that is not a power of two takes the dividing kernels whatever the switch says.  Block orders 0 / 1 / 2 as well."""
    rng = np.random.default_rng(sum(n3))
    try:
        for rg in ([0, 1, 0, 1, 0, 1], [-1, 1, 0, 2, 0, 4], [0, 1, 0, 3, 0, 1]):
            for dtype in (np.float32, np.float64):
                v = (rng.uniform(-1, 1, O.shape(n3)) * 10.0 ** rng.integers(-30, 30)).astype(dtype)
                f = (rng.uniform(-1, 1, O.shape(n3)) * 10.0 ** rng.integers(-30, 30)).astype(dtype)
                for mode in (P.REF_COMPAT, P.CORRECT):
                    want = O.restrict3d(n3, O.residual3d(n3, rg, v, f, mode, dtype=dtype), dtype=dtype)
                    for rcp, xcd, rows in ((1, 1, 4), (0, 1, 2), (1, 0, 2), (1, 2, 4)):
                        ctx.set_param("residual_restrict3d.rcp", rcp)
                        ctx.set_param("residual_restrict3d.xcd", xcd)
                        ctx.set_param("residual_restrict3d.rows", rows)
                        assert bits_equal(P.ops3dxs.residual_restrict(ctx, v, f, n3, rg, mode), want), (rg, dtype, mode, rcp, xcd, rows)
    finally:
        ctx.set_param("residual_restrict3d.rcp", 1)
        ctx.set_param("residual_restrict3d.xcd", 1)
        ctx.set_param("residual_restrict3d.rows", 0)


@pytest.mark.parametrize("layout", ["natural", "xsplit"])
@pytest.mark.parametrize("n3", [(3, 3, 3), (5, 9, 17), (17, 17, 17), (17, 5, 3), (9, 17, 9)])
def test_3d_small_level_one_workgroup_relax(ctx, n3, layout):
    """levels <= 17^3: all sweeps of a Relax call in one workgroup with v, f in LDS == multi-launch path == oracle"""
    ops = OPS3[layout]
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(sum(n3))
    for dtype in (np.float32, np.float64):
        v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
        f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
        for k in (1, 2, 7):
            want = O.relax3d(n3, rg, v, f, k, dtype=dtype)
            for small in (1, 0):
                ctx.set_param("relax3d.small", small)
                assert bits_equal(ops.relax(ctx, v, f, n3, rg, k), want)
    ctx.set_param("relax3d.small", 1)


def test_3d_xsplit_pack_unpack(ctx):
    """device Natural <-> XSplit conversion against the numpy restatement of the layout"""
    rng = np.random.default_rng(11)
    for shape in ((3, 3, 3), (5, 9, 17), (9, 5, 129)):
        for dtype in (np.float32, np.float64):
            a = rng.uniform(-1, 1, shape).astype(dtype)
            assert bits_equal(P.ops3d.pack(ctx, a), P.xs_pack(a))
            assert bits_equal(P.ops3d.unpack(ctx, P.xs_pack(a), shape[-1]), a)
            assert bits_equal(P.xs_unpack(P.xs_pack(a), shape[-1]), a)


@pytest.mark.parametrize("ty,rows,zchunk", [(1, 1, 1), (2, 2, 3), (4, 4, 0), (8, 1, 64), (4, 8, 7), (1, 8, 2), (2, 4, 5)])
def test_3d_xsplit_relax_tuning_knobs_do_not_change_results(ctx, ty, rows, zchunk):
    """waves per block / rows per lane / z-chunk length of the marching smoother are speed knobs only"""
    n3, rg = (65, 33, 41), [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(5)
    v = rng.uniform(-1, 1, O.shape(n3))
    f = rng.uniform(-1, 1, O.shape(n3))
    ctx.set_param("relax3d.ty", ty)
    ctx.set_param("relax3d.rows", rows)
    ctx.set_param("relax3d.zchunk", zchunk)
    try:
        for xcd in (0, 1, 2):
            ctx.set_param("relax3d.xcd", xcd)
            assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, rg, 3), O.relax3d(n3, rg, v, f, 3, dtype=np.float64))
        # time-skewed pass order over z-slabs of every height, including degenerate ones
        for wp in (1, 2, 3, 5, 8, 13, 39, 64):
            ctx.set_param("relax3d.wave_planes", wp)
            for k in (1, 2, 3):
                assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, rg, k), O.relax3d(n3, rg, v, f, k, dtype=np.float64))
    finally:
        ctx.set_param("relax3d.ty", 4)
        ctx.set_param("relax3d.rows", 4)
        ctx.set_param("relax3d.zchunk", 0)
        ctx.set_param("relax3d.xcd", 1)
        ctx.set_param("relax3d.wave_planes", 0)


@pytest.mark.parametrize("code", [1282, 1442, 1242, 1422, 1184, 1424, 3282])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_3d_xsplit_relax_lds_exchange_shapes(ctx, code, dtype):
    """relax3d_xs_pipe_kernel (edge rows / edge lanes handed over through LDS, loads one plane ahead, stores one plane
    behind) == oracle for every workgroup shape of the product build, on sizes where rows, lanes and planes do not fill
    the tile, with short and long z-chunks"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(code)
    ctx.set_param("relax3d.lds", code)
    try:
        for n3 in ((129, 33, 17), (257, 65, 9), (513, 17, 9), (1025, 33, 5), (129, 129, 33)):
            v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
            f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
            for zchunk in (0, 1, 3, 64):
                ctx.set_param("relax3d.zchunk", zchunk)
                assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, rg, 2), O.relax3d(n3, rg, v, f, 2, dtype=dtype)), (n3, zchunk)
    finally:
        ctx.set_param("relax3d.lds", -1)
        ctx.set_param("relax3d.zchunk", 0)


def test_diagnostic_knobs_are_not_in_the_product_build(ctx):
    """the ablation variants of the smoother (wrong results by construction) and the measured-slower A/B kernels exist
    only in `make diag` builds (libmgx_diag.so, tools/); the product library rejects their knobs"""
    for name, value in (("relax3d.ablate", 1), ("relax3d.ablate", 16), ("relax3d.lds", 424), ("relax3d.lds", 12345),
                        ("residual_restrict3d.tyw", 3), ("residual_restrict3d.cr", 7), ("residual_restrict3d.stream", 9)):
        with pytest.raises(P.MgxError) as e:
            ctx.set_param(name, value)
        assert e.value.status == P.MGX_ERR_INVALID


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(9, 9, 9), (17, 33, 9), (129, 17, 33)])
def test_3d_xsplit_interpolate_correct_one_colour(ctx, dtype, n3):
    """mgx3dxs_interpolate_correct_colour: the points of the colour get the reference's correction, the rest is untouched"""
    rng = np.random.default_rng(11)
    cn = P.coarse_size(n3)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
    full = O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3), dtype), c, dtype=dtype), dtype=dtype)
    z, y, x = np.meshgrid(*[np.arange(k) for k in n3[::-1]], indexing="ij")
    par = (x + y + z) & 1
    for colour in (0, 1):
        got = P.ops3dxs.interpolate_correct_colour(ctx, v, n3, c, colour)
        assert bits_equal(got, np.where(par == colour, full, v))
    assert bits_equal(P.ops3dxs.interpolate_correct_colour(ctx, v, n3, c, -1), full)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(257, 129, 33), (513, 129, 17), (257, 257, 17), (1025, 129, 17), (257, 65, 9), (513, 513, 33), (33, 17, 65),
                                (9, 9, 9), (129, 129, 33)])
def test_3d_xsplit_interpolate_correct_relax_fused(ctx, dtype, n3):
    """mgx3dxs_interpolate_correct_relax == Relax(ApplyCorrection(Interpolate)) (N3/MultiGrid3D.cpp:638-645).  On wide
    levels the first red pass reads the black points through the correction (values of tile-edge cells are corrected in
    place beforehand, everything else on the fly from coarse values held in registers); random data, so a correction
    that is missing, doubled or taken from the wrong coarse cell anywhere shows; all run lengths of the plane march"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(sum(n3))
    cn = P.coarse_size(n3)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
    corrected = O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3), dtype), c, dtype=dtype), dtype=dtype)
    try:
        for fuse in (1, 0):
            ctx.set_param("relax3d.corr_fuse", fuse)
            for zchunk in (0, 3, 8, 64):
                ctx.set_param("relax3d.zchunk", zchunk)
                for k in (1, 2):
                    want = O.relax3d(n3, rg, corrected, f, k, dtype=dtype)
                    assert bits_equal(P.ops3dxs.interpolate_correct_relax(ctx, v, f, n3, rg, c, k), want), (fuse, zchunk, k)
    finally:
        ctx.set_param("relax3d.corr_fuse", 1)
        ctx.set_param("relax3d.zchunk", 0)


@pytest.mark.parametrize("n3", [(513, 129, 17), (513, 257, 33), (1025, 129, 17), (2049, 129, 17)])
def test_3d_xsplit_correcting_pass_two_pairs_per_lane(ctx, n3):
    """fp32, rows of >= 513 points: the correcting red pass runs as relax3d_xs_pipe_v2_kernel<float,...,2> (tiles of 256
    pairs: set P has a cell column every 256 pairs, a lane interpolates for both of its pairs); "relax3d.corr_v2" = 0 is the
    one-pair-per-lane kernel.  Both against the oracle"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(sum(n3) + 1)
    cn = P.coarse_size(n3)
    v, f = (rng.uniform(-1, 1, O.shape(n3)).astype(np.float32) for _ in range(2))
    c = rng.uniform(-1, 1, O.shape(cn)).astype(np.float32)
    corrected = O.correct3d(n3, v, O.interpolate3d(n3, np.zeros(O.shape(n3), np.float32), c, dtype=np.float32), dtype=np.float32)
    want = O.relax3d(n3, rg, corrected, f, 1, dtype=np.float32)
    try:
        for on, name in ((1, "relax3d_xs_pipe_v2_kernel<float"), (0, "relax3d_xs_pipe_kernel<float")):
            ctx.set_param("relax3d.corr_v2", on)
            for zchunk in (0, 2, 5):
                ctx.set_param("relax3d.zchunk", zchunk)
                assert bits_equal(P.ops3dxs.interpolate_correct_relax(ctx, v, f, n3, rg, c, 1), want), (on, zchunk)
                assert ctx.last_corr_kernel().startswith(name), (on, ctx.last_corr_kernel())
    finally:
        ctx.set_param("relax3d.corr_v2", 1)
        ctx.set_param("relax3d.zchunk", 0)


@pytest.mark.parametrize("layout", ["natural", "xsplit"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(3, 3, 3), (9, 17, 9), (33, 17, 65), (129, 65, 17), (257, 129, 33), (513, 129, 17)])
def test_3d_relax_from_zero(ctx, layout, dtype, n3):
    """mgx3d[xs]_relax_from_zero == setToValue(v, 0, true) + Relax (N3/MultiGrid3D.cpp:634, :626).  With rim_is_zero the
    first red pass does not read v and nothing is filled: v's interior is handed in as garbage (NaN) to prove it, only
    its boundary is zero; without it v is garbage everywhere and is zero-filled first"""
    ops = OPS3[layout]
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(sum(n3))
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    zeros = np.zeros(O.shape(n3), dtype)
    garbage = np.full(O.shape(n3), np.nan, dtype)
    rim0 = garbage.copy()
    rim0[0], rim0[-1], rim0[:, 0], rim0[:, -1], rim0[:, :, 0], rim0[:, :, -1] = 0, 0, 0, 0, 0, 0
    try:
        for zero_first in (1, 0):
            ctx.set_param("relax3d.zero_first", zero_first)
            for k in (0, 1, 2):
                want = O.relax3d(n3, rg, zeros, f, k, dtype=dtype)
                assert bits_equal(ops.relax_from_zero(ctx, garbage, f, n3, rg, k, False), want), (zero_first, k)
                assert bits_equal(ops.relax_from_zero(ctx, rim0, f, n3, rg, k, True), want), (zero_first, k, "rim")
    finally:
        ctx.set_param("relax3d.zero_first", 1)


@pytest.mark.parametrize("n3", [(513, 129, 17), (1025, 129, 17), (513, 257, 9), (2049, 129, 9), (513, 513, 33)])
def test_3d_xsplit_relax_fp32_two_pairs_per_lane(ctx, n3):
    """relax3d_xs_pipe_v2_kernel (fp32 on wide levels: a lane owns two x-pairs, 8-byte loads) == oracle == the one-pair
    kernel, on rows that do and do not fill the 256-pair tiles, with every run length of the plane march"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(sum(n3))
    v = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)
    want = O.relax3d(n3, rg, v, f, 2, dtype=np.float32)
    try:
        for v2 in (1, 0):
            ctx.set_param("relax3d.v2", v2)
            for zchunk in (0, 1, 3, 8, 64):
                ctx.set_param("relax3d.zchunk", zchunk)
                assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, rg, 2), want), (v2, zchunk)
                if v2 and zchunk == 0 and n3[2] - 2 >= 8:
                    assert "v2" in ctx.last_relax_kernel()
    finally:
        ctx.set_param("relax3d.v2", 1)
        ctx.set_param("relax3d.zchunk", 0)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n3", [(257, 129, 33), (513, 129, 17), (257, 257, 12 + 5), (1025, 129, 9)])
def test_3d_xsplit_relax_default_kernel_choice_large_rows(ctx, dtype, n3):
    """default parameters on levels wide enough for the automatic choice of the pipelined LDS-exchange smoother"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(3)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    assert bits_equal(P.ops3dxs.relax(ctx, v, f, n3, rg, 2), O.relax3d(n3, rg, v, f, 2, dtype=dtype))


def test_3d_size_violations_return_status(ctx):
    v = np.zeros((9, 9, 9))
    with pytest.raises(P.MgxError) as e:  # reference: assert(csize == (fsize-1)/2+1)  N3/MultiGrid3D.cpp:60-62
        P.ops3d.restrict(ctx, v, [9, 9, 9], cn=[4, 5, 5])
    assert e.value.status == P.MGX_ERR_SIZE
    with pytest.raises(P.MgxError) as e:  # N3/MultiGrid3D.cpp:660-662
        P.ops3d.apply_correction(ctx, v, [9, 9, 9], v, en=[9, 9, 5])
    assert e.value.status == P.MGX_ERR_SIZE
    with pytest.raises(P.MgxError) as e:  # N3/Grid3D.cpp:13
        P.MultiGrid3D(ctx, [10, 10, 10], R3)
    assert e.value.status == P.MGX_ERR_SIZE


# ------------------------------------------------------------------ 3D cycles
@pytest.mark.parametrize("name", ["3d_n9_fmg122", "3d_n17_fmg122", "3d_n33_vcycle22", "3d_n65_vcycle22",
                                  "3d_n129_relax10", "3d_n257_vcycle22_6lev", "3d_n257_relax4", "3d_n513_vcycle22_9lev"])
@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("layout", ["natural", "xsplit"])
def test_3d_reference_known_answers_f32(ctx, known_answers, name, fuse, layout):
    """analytic InitF on the device (host sin tables) + cycles == the unmodified reference, bit for bit;
    3d_n257_vcycle22_6lev is BASELINE.json configs[2] (in the reference's own fp32)."""
    ka = known_answers[name]
    n = ka["n"]
    if fuse is False and n > 129:
        pytest.skip("unfused path covered at smaller sizes")
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float32, nlevels=ka.get("nlevels", 0), fuse=fuse, layout=layout)
    if ka["mode"] == 0:
        mg.VCycle(0, ka["v1"], ka["v2"])
    else:
        mg.FullMultiGridVCycle(0, ka["v0"], ka["v1"], ka["v2"])
    v = mg.download_v(0)
    assert O.fnv(v) == ka["hash"]
    assert float(v[n // 2, n // 2, n // 2]) == ka["centre"]
    mg.close()


@pytest.mark.parametrize("n,nlev", [(33, 0), (65, 4), (129, 0)])
@pytest.mark.parametrize("mode", [P.REF_COMPAT, P.CORRECT])
@pytest.mark.parametrize("layout", ["natural", "xsplit"])
def test_3d_cycles_f64_vs_oracle(ctx, n, nlev, mode, layout):
    for fmg in (False, True):
        mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64, nlevels=nlev, residual_mode=mode, layout=layout)
        if fmg:
            mg.FullMultiGridVCycle(0, 1, 2, 2)
        else:
            mg.VCycle(0, 2, 2)
            mg.VCycle(0, 2, 2)
        want = O.cycle3d([n] * 3, R3, nlevels=nlev, mode=int(fmg), v0=1, v1=2, v2=2, reps=2, residual_mode=mode,
                         dtype=np.float64)
        assert_f64(mg.download_v(0), want)
        mg.close()


@pytest.mark.parametrize("small", [1, 0])
@pytest.mark.parametrize("layout", ["natural", "xsplit"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_3d_one_workgroup_tail_of_the_cycle(ctx, small, layout, dtype):
    """levels of at most 17^3: the rest of the V-cycle (way down and way up) in ONE launch with every level in LDS
    ("relax3d.small" = 1, the default) == one launch per operator ("relax3d.small" = 0) == oracle; hierarchies that end
    in the tail, start inside it, are anisotropic, stop early (numGrids), use 0 sweeps, both residual modes"""
    rg = [-1, 1, 0, 2, 0.5, 3]
    cases = [((33, 33, 33), 0, 2, 2, P.REF_COMPAT), ((65, 33, 17), 0, 1, 3, P.CORRECT), ((17, 17, 17), 0, 2, 2, P.REF_COMPAT),
             ((9, 17, 9), 0, 0, 2, P.CORRECT), ((5, 5, 5), 0, 2, 0, P.REF_COMPAT), ((3, 3, 3), 0, 3, 1, P.REF_COMPAT),
             ((33, 33, 33), 2, 2, 2, P.REF_COMPAT), ((33, 17, 33), 3, 1, 1, P.CORRECT), ((17, 17, 17), 1, 2, 2, P.REF_COMPAT)]
    ctx.set_param("relax3d.small", small)
    try:
        for n3, nlev, v1, v2, mode in cases:
            rng = np.random.default_rng(sum(n3) + nlev)
            v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
            f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
            mg = P.MultiGrid3D(ctx, n3, rg, dtype, nlevels=nlev, residual_mode=mode, layout=layout)
            mg.upload_v(0, v)
            mg.upload_f(0, f)
            mg.VCycle(0, v1, v2)
            mg.VCycle(0, v1, v2)
            want = O.cycle3d(n3, rg, nlevels=nlev, mode=0, v1=v1, v2=v2, reps=2, v=v, f=f, residual_mode=mode, dtype=dtype)
            assert bits_equal(mg.download_v(0), want), (n3, nlev, v1, v2, mode)
            mg.close()
            mg = P.MultiGrid3D(ctx, n3, rg, dtype, nlevels=nlev, residual_mode=mode, layout=layout)
            mg.upload_v(0, v)
            mg.upload_f(0, f)
            mg.FullMultiGridVCycle(0, 2, v1, v2)
            want = O.cycle3d(n3, rg, nlevels=nlev, mode=1, v0=2, v1=v1, v2=v2, v=v, f=f, residual_mode=mode, dtype=dtype)
            assert bits_equal(mg.download_v(0), want), ("fmg", n3, nlev, v1, v2, mode)
            mg.close()
    finally:
        ctx.set_param("relax3d.small", 1)


def test_3d_baseline_config2_257_f64(ctx):
    """BASELINE.json configs[2]: 3D Poisson 256^3 (257 points/axis), 6-level V-cycle, fp64."""
    mg = P.MultiGrid3D(ctx, [257] * 3, R3, np.float64, nlevels=6)
    mg.VCycle(0, 2, 2)
    want = O.cycle3d([257] * 3, R3, nlevels=6, mode=0, dtype=np.float64)
    assert_f64(mg.download_v(0), want)
    # residual norm (an addition; parity unpinned by the reference): against numpy on the oracle's residual
    r = O.residual3d([257] * 3, R3, want, O.init3d([257] * 3, R3, 0, np.float64)[1], dtype=np.float64)
    assert abs(mg.ResidualNorm(0) - np.linalg.norm(r.ravel())) <= 1e-9 * np.linalg.norm(r.ravel())
    mg.close()


def test_3d_correct_mode_converges_like_textbook_multigrid(ctx):
    """CORRECT residual, FMG(1,2,2) at 129^3: rel-L2 error vs analytic 1.15e-4"""
    n = 129
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64, residual_mode=P.CORRECT)
    mg.FullMultiGridVCycle(0, 1, 2, 2)
    v = mg.download_v(0)
    s = np.sin(np.pi * np.linspace(0, 1, n))
    u = s[None, None, :] * s[None, :, None] * s[:, None, None]
    assert rel_l2(v, u) < 2e-4
    mg.close()


def test_3d_full_size_513_sweep_and_properties(ctx):
    """BASELINE.json configs[3] size (513 points/axis, fp64): one smoother sweep against the oracle on the
    whole array, plus size-independent properties: boundary untouched, idempotent set, and
    restrict(interpolate(c)) == c on coarse-aligned points is NOT expected (full weighting), so the
    property used is interpolate_correct(v, 0) == v bit for bit."""
    n = 513
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64, nlevels=1)
    f = mg.download_f(0)
    mg.Relax(0, 1)
    got = mg.download_v(0)
    want = O.relax3d([n] * 3, R3, np.zeros_like(f), f, 1, dtype=np.float64)
    assert_f64(got, want)
    for face in (got[0], got[-1], got[:, 0], got[:, -1], got[:, :, 0], got[:, :, -1]):
        assert not face.any()
    del want, f
    mg.close()
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64, nlevels=2)
    mg.upload_v(0, got)
    mg.setToValue_v(1, 0.0, True)
    mg.interpolate_correct(0)
    assert bits_equal(mg.download_v(0), got)
    mg.close()


@pytest.mark.timeout(600)
def test_3d_bench_workload_513_vcycle_bit_exact(ctx):
    """the workload of bench.py itself -- BASELINE.json configs[3]: 513 points per axis, fp64, native 9 levels, analytic
    RHS, v = 0, one V(2,2) cycle with default parameters (pipelined smoother, fused operators, black-only correction)
    -- against the oracle on every one of the 135 M points (about 15 s of single-threaded CPU)"""
    n = 513
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64)
    assert mg.numGrids == 9
    mg.VCycle(0, 2, 2)
    got = mg.download_v(0)
    mg.close()
    want = O.cycle3d([n] * 3, R3, mode=0, v1=2, v2=2, reps=1, dtype=np.float64)
    assert bits_equal(got, want)


# ------------------------------------------------------------------ 2D
@pytest.mark.parametrize("n", [9, 17, 33])
def test_2d_ops_f32_vs_reference_fixtures(ctx, n):
    g = load_golden("ops2d_n%d.npz" % n)
    n2, rg, A, alfa = g["n"].tolist(), g["range"].tolist(), g["A"].tolist(), int(g["alfa"])
    v, f, c = g["v"], g["f"], g["c"]
    assert bits_equal(P.ops2d.relax(ctx, v, f, n2, rg, A, alfa, 1), g["relax1"])
    assert bits_equal(P.ops2d.relax(ctx, v, f, n2, rg, A, alfa, 3), g["relax3"])
    assert bits_equal(P.ops2d.residual(ctx, v, f, n2, rg, A, alfa), g["residual"])
    assert bits_equal(P.ops2d.restrict(ctx, v, n2), g["restrict"])
    assert bits_equal(P.ops2d.interpolate(ctx, v, n2, c), g["interpolate"])
    assert bits_equal(P.ops2d.apply_correction(ctx, v, n2, f), g["correct"])
    assert bits_equal(P.ops2d.set(ctx, v, n2, 2.5, False), g["set_interior"])
    assert bits_equal(P.ops2d.set(ctx, v, n2, 2.5, True), g["set_all"])
    assert bits_equal(P.solve2d(ctx, v, f, rg, A, alfa, ncycles=1), g["vcycle22"])
    assert bits_equal(P.solve2d(ctx, v, f, rg, A, alfa, fmg=True, v0=1), g["fmg122"])


@pytest.mark.parametrize("n2", [(9, 9), (65, 17), (129, 257)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_ops_vs_oracle_random(ctx, n2, dtype):
    rng = np.random.default_rng(sum(n2))
    rg = [0, 20, -3, 20]
    v = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(O.csize(n2))).astype(dtype)
    for k in (1, 3):
        assert bits_equal(P.ops2d.relax(ctx, v, f, n2, rg, A2, 2, k), O.relax2d(n2, rg, A2, 2, v, f, k, dtype=dtype))
    assert bits_equal(P.ops2d.residual(ctx, v, f, n2, rg, A2, 2), O.residual2d(n2, rg, A2, 2, v, f, dtype=dtype))
    assert bits_equal(P.ops2d.restrict(ctx, v, n2), O.restrict2d(n2, v, dtype=dtype))
    assert bits_equal(P.ops2d.interpolate(ctx, v, n2, c), O.interpolate2d(n2, v, c, dtype=dtype))
    assert bits_equal(P.ops2d.apply_correction(ctx, v, n2, f), O.correct2d(n2, v, f, dtype=dtype))
    for b in (0, 1):
        assert bits_equal(P.ops2d.set(ctx, v, n2, 3.5, b), O.set2d(n2, v, 3.5, b, dtype=dtype))


@pytest.mark.parametrize("name", ["2d_n33_fmg122", "2d_n129_fmg122", "2d_n257_fmg_1_500_500", "2d_n1025_vcycle22_7lev",
                                  "2d_n1025_relax20"])
def test_2d_reference_known_answers_f32(ctx, known_answers, name):
    """2d_n1025_vcycle22_7lev is BASELINE.json configs[1] in the reference's own fp32."""
    ka = known_answers[name]
    n = ka["n"]
    mg = P.MultiGrid2D(ctx, [n] * 2, [0, 1, 0, 1], A2, 2, np.float32, nlevels=ka.get("nlevels", 0))
    if ka["mode"] == 0:
        mg.VCycle(0, ka["v1"], ka["v2"])
    else:
        mg.FullMultiGridVCycle(0, ka["v0"], ka["v1"], ka["v2"])
    v = mg.download_v(0)
    assert O.fnv(v) == ka["hash"]
    assert float(v[n // 2, n // 2]) == ka["centre"]
    mg.close()


@pytest.mark.parametrize("n2", [(9, 9), (33, 17), (129, 65), (17, 257)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_fused_operators_vs_oracle(ctx, n2, dtype):
    """mgx2d_residual_restrict == Restrict(CalculateResidual), mgx2d_interpolate_correct == ApplyCorrection(Interpolate)"""
    rng = np.random.default_rng(sum(n2))
    rg = [-1, 2, 0.5, 3]
    cn = P.coarse_size(n2)
    v = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
    want = O.restrict2d(n2, O.residual2d(n2, rg, A2, 2, v, f, dtype=dtype), dtype=dtype)
    assert bits_equal(P.ops2d.residual_restrict(ctx, v, f, n2, rg, A2, 2), want)
    want = O.correct2d(n2, v, O.interpolate2d(n2, np.zeros(O.shape(n2), dtype), c, dtype=dtype), dtype=dtype)
    assert bits_equal(P.ops2d.interpolate_correct(ctx, v, n2, c), want)


@pytest.mark.parametrize("n2", [(3, 3), (9, 9), (17, 33), (65, 65), (129, 65), (33, 257), (257, 257), (513, 129), (1025, 513)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_cache_resident_kernels_vs_oracle(ctx, n2, dtype):
    """the LDS-tiled multi-sweep kernels of the 2D cycle == the reference calls they replace, on sizes that do and do not
    fill the tiles (16 / 32 / 64 points), 0 ... 4 sweeps per launch:
      relax_residual_restrict   = Relax(k), then Restrict(CalculateResidual(.))     N2/MultiGrid2D.cpp:317-323
      interpolate_correct_relax = Relax(ApplyCorrection(Interpolate(.)), k)         N2/MultiGrid2D.cpp:333-338"""
    rng = np.random.default_rng(sum(n2))
    rg = [-1, 2, 0.5, 3]
    cn = P.coarse_size(n2)
    v = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
    c = rng.uniform(-1, 1, O.shape(cn)).astype(dtype)
    for k in (0, 1, 2, 4):
        vk = O.relax2d(n2, rg, A2, 2, v, f, k, dtype=dtype)
        got_v, got_c = P.ops2d.relax_residual_restrict(ctx, v, f, n2, rg, A2, 2, k)
        assert bits_equal(got_v, vk), k
        assert bits_equal(got_c, O.restrict2d(n2, O.residual2d(n2, rg, A2, 2, vk, f, dtype=dtype), dtype=dtype)), k
        got_v, _ = P.ops2d.relax_residual_restrict(ctx, v, f, n2, rg, A2, 2, k, restrict=False)
        assert bits_equal(got_v, vk), k
        z = np.zeros_like(v)  # v_zero: the input is not read and counts as the zeroed coarse error
        zk = O.relax2d(n2, rg, A2, 2, z, f, k, dtype=dtype)
        got_v, got_c = P.ops2d.relax_residual_restrict(ctx, np.full_like(v, np.nan), f, n2, rg, A2, 2, k, v_zero=True)
        assert bits_equal(got_v, zk), k
        assert bits_equal(got_c, O.restrict2d(n2, O.residual2d(n2, rg, A2, 2, zk, f, dtype=dtype), dtype=dtype)), k
        want = O.relax2d(n2, rg, A2, 2, O.correct2d(n2, v, O.interpolate2d(n2, np.zeros(O.shape(n2), dtype), c, dtype=dtype), dtype=dtype),
                         f, k, dtype=dtype)
        assert bits_equal(P.ops2d.interpolate_correct_relax(ctx, v, f, n2, rg, A2, 2, c, k), want), k
    with pytest.raises(P.MgxError) as e:  # the tile halo is sized for 4 sweeps
        P.ops2d.relax_residual_restrict(ctx, v, f, n2, rg, A2, 2, 5)
    assert e.value.status == P.MGX_ERR_INVALID


@pytest.mark.parametrize("fuse,params", [(2, {}), (1, {}), (0, {}), (2, {"cycle2d.tail_points": 4225}), (2, {"cycle2d.tail_points": 0}),
                                         (2, {"cycle2d.tile": 32}), (2, {"cycle2d.tile": 64, "cycle2d.tail_points": 289})])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_cycle_paths_agree_with_oracle(ctx, fuse, params, dtype):
    """the three ways the host layer can run VCycle (cache-resident kernels / fused operators / one launch per reference
    call) on hierarchies that end in the one-workgroup tail, start inside it, or end on a level too big for it; the
    cache-resident path also with every tile size and with the tail starting at 65^2, at 17^2 and not at all"""
    rg = [0, 1, 0, 2]
    for k_, v_ in params.items():
        ctx.set_param(k_, v_)
    try:
        _cycle_paths_2d(ctx, fuse, dtype, rg)
    finally:
        ctx.set_param("cycle2d.tile", 0)
        ctx.set_param("cycle2d.tail_points", 33 * 33)


def _cycle_paths_2d(ctx, fuse, dtype, rg):
    cases = [((257, 257), 0, 2, 2), ((129, 65), 0, 1, 3), ((65, 65), 0, 2, 2), ((33, 17), 0, 0, 2), ((9, 9), 0, 2, 0),
             ((257, 129), 2, 2, 2), ((513, 513), 1, 1, 1), ((129, 129), 3, 2, 1), ((129, 129), 0, 4, 4)]
    for n2, nlev, v1, v2 in cases:
        rng = np.random.default_rng(sum(n2) + nlev)
        v = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
        f = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
        mg = P.MultiGrid2D(ctx, n2, rg, A2, 2, dtype, nlevels=nlev, fuse=fuse)
        mg.upload_v(0, v)
        mg.upload_f(0, f)
        for _ in range(2):
            mg.VCycle(0, v1, v2)
        want = O.cycle2d(n2, rg, A2, 2, nlevels=nlev, mode=0, v1=v1, v2=v2, reps=2, v=v, f=f, dtype=dtype)
        assert bits_equal(mg.download_v(0), want), (n2, nlev, v1, v2)
        mg.close()
        mg = P.MultiGrid2D(ctx, n2, rg, A2, 2, dtype, nlevels=nlev, fuse=fuse)  # FMG on a fresh hierarchy: VCycle from every level
        mg.upload_v(0, v)
        mg.upload_f(0, f)
        mg.FullMultiGridVCycle(0, 2, v1, v2)
        want = O.cycle2d(n2, rg, A2, 2, nlevels=nlev, mode=1, v0=2, v1=v1, v2=v2, f=f, v=v, dtype=dtype)
        assert bits_equal(mg.download_v(0), want), ("fmg", n2, nlev, v1, v2)
        mg.close()


def test_2d_unfused_cycle_path_still_matches(ctx):
    mg = P.MultiGrid2D(ctx, [129] * 2, [0, 1, 0, 1], A2, 2, np.float64, fuse=False)
    mg.FullMultiGridVCycle(0, 1, 2, 2)
    assert_f64(mg.download_v(0), O.cycle2d([129] * 2, [0, 1, 0, 1], A2, 2, mode=1, v0=1, v1=2, v2=2, dtype=np.float64))
    mg.close()


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_2d_long_relax_calls_run_four_sweeps_per_launch(ctx, dtype):
    """Relax(g, ncycles) on a level that does not fit one workgroup: up to four sweeps per launch through the tiled kernel,
    in an even number of out-of-place launches (the result is back in d_v, no pointer changes hands) == oracle, for sweep
    counts around the chunking's edges; the per-colour path (fuse = 1) too; a graph-captured cycle with v1 = v2 = 7"""
    rg = [0, 20, 0, 20]
    for n2 in ((129, 129), (257, 129), (129, 513)):
        rng = np.random.default_rng(sum(n2))
        v = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
        f = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
        for fuse in (2, 1):
            mg = P.MultiGrid2D(ctx, n2, rg, A2, 2, dtype, fuse=fuse)
            ptr = mg.grid(0).d_v
            for sweeps in (2, 3, 4, 5, 8, 9, 13, 30) if fuse == 2 else (5,):
                mg.upload_v(0, v)
                mg.upload_f(0, f)
                mg.Relax(0, sweeps)
                assert mg.grid(0).d_v == ptr
                assert bits_equal(mg.download_v(0), O.relax2d(n2, rg, A2, 2, v, f, sweeps, dtype=dtype)), (n2, fuse, sweeps)
            mg.close()
    n2 = (257, 257)
    want = O.cycle2d(n2, rg, A2, 2, mode=0, v1=7, v2=7, reps=3, dtype=dtype)
    mg = P.MultiGrid2D(ctx, n2, rg, A2, 2, dtype)
    mg.use_graph = True
    for _ in range(3):
        mg.VCycle(0, 7, 7)
    assert bits_equal(mg.download_v(0), want)
    mg.close()


def test_2d_baseline_config1_1025_f64(ctx):
    """BASELINE.json configs[1]: 2D Lyapunov 1024x1024 (1025 points/axis), 7-level V-cycle, fp64."""
    mg = P.MultiGrid2D(ctx, [1025] * 2, [0, 1, 0, 1], A2, 2, np.float64, nlevels=7)
    mg.VCycle(0, 2, 2)
    assert_f64(mg.download_v(0), O.cycle2d([1025] * 2, [0, 1, 0, 1], A2, 2, nlevels=7, mode=0, dtype=np.float64))
    mg.close()
    mg = P.MultiGrid2D(ctx, [257] * 2, [0, 20, 0, 20], A2, 2, np.float64)
    mg.FullMultiGridVCycle(0, 1, 50, 50)
    assert_f64(mg.download_v(0), O.cycle2d([257] * 2, [0, 20, 0, 20], A2, 2, mode=1, v0=1, v1=50, v2=50, dtype=np.float64))
    mg.close()


@pytest.mark.parametrize("n", [1025, 4097])
def test_2d_published_workload_full_parameters_known_answer(ctx, n):
    """the reference's published 2D workload with its own parameters (thesis Fig. 4.2 = BASELINE.md section 1; N2/LyapunovSolver.cpp:13-31
    on [0, 20]^2: FMG with 2 V-cycles per level, 500 + 500 sweeps per visit, fp32) against the hash of the oracle's result
    (oracle/gen_known_f64.py thesis2d: minutes of CPU at 4097^2; restatement<float> = the compiled reference's bits)"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers_f64.json")) as fh:
        ka = json.load(fh)["2d_n%d_fmg_2_500_500_f32" % n]
    mg = P.MultiGrid2D(ctx, [n] * 2, [0, 20, 0, 20], A2, 2, np.float32)
    assert mg.numGrids == ka["nlevels"]
    mg.FullMultiGridVCycle(0, 2, 500, 500)
    got = mg.download_v(0)
    mg.close()
    assert O.fnv(got) == ka["fnv"] and float(got[n // 2, n // 2]) == ka["centre"]


@pytest.mark.parametrize("layout", ["xsplit", "natural"])
def test_3d_coarse_rhs_rim_is_zero_after_every_cycle(ctx, layout):
    """the cycle re-zeroes the boundary entries of a coarse level's f only when something wrote there since the last
    residual+restrict (flag f_rim_zero); whatever the history, after a cycle they are 0 like in the reference"""
    n = 33
    mg = P.MultiGrid3D(ctx, [n] * 3, [-1, 1, 0, 2, 0.5, 3], np.float64, layout=layout)

    def rim_is_zero(level):
        f = mg.download_f(level)
        inner = f[1:-1, 1:-1, 1:-1].copy()
        f[1:-1, 1:-1, 1:-1] = 0
        return not f.any() and inner.any()

    mg.VCycle(0, 2, 2)
    assert rim_is_zero(1) and rim_is_zero(2)
    mg.VCycle(0, 2, 2)  # second cycle: the keep-rim form of residual+restrict
    assert rim_is_zero(1) and rim_is_zero(2)
    junk = np.random.default_rng(0).uniform(1, 2, (17, 17, 17))
    mg.upload_f(1, junk)  # non-zero boundary written from outside
    mg.VCycle(0, 2, 2)
    assert rim_is_zero(1)
    mg.FullMultiGridVCycle(0, 1, 2, 2)  # FMG's Restrict(f) injects the fine boundary, the cycles zero it again
    assert rim_is_zero(1) and rim_is_zero(2)
    want = None
    mg.close()
    # and the values of v are those of the oracle after the same sequence of calls on fresh hierarchies (fused paths)
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64, layout=layout)
    for _ in range(3):
        mg.VCycle(0, 2, 2)
    want = O.cycle3d([n] * 3, R3, mode=0, v1=2, v2=2, reps=3, dtype=np.float64)
    assert bits_equal(mg.download_v(0), want)
    mg.close()


def test_hip_graph_replay_of_cycles_2d_and_3d(ctx):
    """use_graph: VCycle captured into a HIP graph on first use and replayed; re-captured when its arguments change;
    FMG (one graph per starting level).  Bit-identical to the launch-by-launch path / the oracle."""
    mg = P.MultiGrid2D(ctx, [1025] * 2, [0, 1, 0, 1], A2, 2, np.float64, nlevels=7)
    mg.use_graph = True
    for _ in range(3):
        mg.VCycle(0, 2, 2)
    mg.VCycle(0, 1, 3)  # other arguments: new capture
    mg.VCycle(0, 2, 2)
    ref = P.MultiGrid2D(ctx, [1025] * 2, [0, 1, 0, 1], A2, 2, np.float64, nlevels=7)
    for _ in range(3):
        ref.VCycle(0, 2, 2)
    ref.VCycle(0, 1, 3)
    ref.VCycle(0, 2, 2)
    assert bits_equal(mg.download_v(0), ref.download_v(0))
    mg.close()
    ref.close()
    mg = P.MultiGrid2D(ctx, [257] * 2, [0, 20, 0, 20], A2, 2, np.float64)
    mg.use_graph = True
    mg.FullMultiGridVCycle(0, 1, 50, 50)
    assert_f64(mg.download_v(0), O.cycle2d([257] * 2, [0, 20, 0, 20], A2, 2, mode=1, v0=1, v1=50, v2=50, dtype=np.float64))
    mg.close()
    for layout in ("xsplit", "natural"):
        mg = P.MultiGrid3D(ctx, [129] * 3, R3, np.float64, residual_mode=P.CORRECT, layout=layout)
        mg.use_graph = True
        mg.FullMultiGridVCycle(0, 2, 2, 2)
        want = O.cycle3d([129] * 3, R3, mode=1, v0=2, v1=2, v2=2, residual_mode=P.CORRECT, dtype=np.float64)
        assert_f64(mg.download_v(0), want)
        mg.numGrids = 4  # public field changed: the key changes, new capture
        mg.VCycle(0, 2, 1)
        mg.close()
    mg = P.MultiGrid3D(ctx, [65] * 3, R3, np.float32)
    mg.use_graph = True
    mg.VCycle(0, 2, 2)
    mg.VCycle(0, 2, 2)
    assert bits_equal(mg.download_v(0), O.cycle3d([65] * 3, R3, mode=0, v1=2, v2=2, reps=2, dtype=np.float32))
    mg.close()


def test_norm2_wave_reduction(ctx):
    rng = np.random.default_rng(0)
    for cnt in (1, 63, 64, 65, 1000003):
        x = rng.uniform(-1, 1, cnt)
        got = P.ops3d.norm2(ctx, x)
        assert abs(got - float(np.dot(x, x))) <= 1e-12 * cnt


# ------------------------------------------------------------------ additions without a reference counterpart
@pytest.mark.parametrize("layout", ["natural", "xsplit"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_weighted_jacobi_3d_vs_oracle(ctx, layout, dtype):
    """weighted Jacobi is named by north_star but absent from the reference: parity unpinned; the HIP kernel must
    equal the oracle's restatement of the same expression bit for bit"""
    ops = OPS3[layout]
    n3, rg = (33, 17, 9), [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(9)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    om = float(dtype(2) / dtype(3))
    for k in (1, 2, 5):
        assert bits_equal(ops.jacobi(ctx, v, f, n3, rg, om, k), O.jacobi3d(n3, rg, v, f, om, k, dtype=dtype))


def test_weighted_jacobi_2d_vs_oracle(ctx):
    n2, rg = (65, 17), [0, 20, -3, 20]
    rng = np.random.default_rng(10)
    for dtype in (np.float32, np.float64):
        v = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
        f = rng.uniform(-1, 1, O.shape(n2)).astype(dtype)
        om = float(dtype(0.8))
        for k in (1, 4):
            assert bits_equal(P.ops2d.jacobi(ctx, v, f, n2, rg, A2, 2, om, k), O.jacobi2d(n2, rg, A2, 2, v, f, om, k, dtype=dtype))


def test_jacobi_vcycle_converges_and_diff_stats(ctx):
    """V-cycles with the weighted-Jacobi smoother (CORRECT residual) reduce the error like multigrid should, and
    DiffStats (Grid3D::PrintDiff as a device reduction) agrees with numpy on the downloaded field"""
    n = 65
    mg = P.MultiGrid3D(ctx, [n] * 3, R3, np.float64, residual_mode=P.CORRECT)
    mg.set_smoother("jacobi", 6.0 / 7.0)
    errs = []
    for _ in range(6):
        mg.VCycle(0, 3, 3)
        errs.append(mg.DiffStats(0)[2])
    assert errs[-1] < 2e-3 and errs[-1] < errs[0]
    v = mg.download_v(0)
    t = np.float64
    x = np.array([t(t(0) + i * (t(1) / t(n - 1))) for i in range(n)])
    s = np.sin(np.pi * x)
    u = s[None, None, :] * s[None, :, None] * s[:, None, None]
    d = u - v
    mean_abs, max_abs, rel_l2 = mg.DiffStats(0)
    assert abs(mean_abs - np.abs(d).mean()) <= 1e-12
    assert abs(max_abs - np.abs(d).max()) <= 1e-12
    assert abs(rel_l2 - np.linalg.norm(d.ravel()) / np.linalg.norm(u.ravel())) <= 1e-10
    mg.close()


def test_3d_solve_from_zero_is_the_reference_driver(ctx, known_answers):
    """mg3d_solve_from_zero with rhs = NULL: construct, InitF on the device, FMG, download -- N3/Poisson3DSolver.cpp:6-51
    with one transfer; fp32 against the compiled reference's known answer, fp64 with an uploaded rhs against the oracle"""
    got = P.solve3d_from_zero(ctx, [17] * 3, R3, np.float32, fmg=True, v0=1, v1=2, v2=2)
    assert O.fnv(got) == known_answers["3d_n17_fmg122"]["hash"]
    got = P.solve3d_from_zero(ctx, [33] * 3, R3, np.float32, ncycles=1)
    assert O.fnv(got) == known_answers["3d_n33_vcycle22"]["hash"]
    n3, rg = [33, 17, 65], [-1, 1, 0, 2, 0.5, 3]
    f = np.random.default_rng(3).uniform(-1, 1, O.shape(n3))
    got = P.solve3d_from_zero(ctx, n3, rg, np.float64, rhs=f, ncycles=2)
    assert_f64(got, O.cycle3d(n3, rg, mode=0, v1=2, v2=2, reps=2, f=f, dtype=np.float64))


def test_2d_initv_on_the_device(ctx):
    """Grid2D::InitV (N2/Grid2D.cpp:50-68) as a kernel: fp32 against the reference's own values, every level, fp64 and an
    anisotropic box with negative coordinates against the oracle"""
    g = load_golden("init2d_n33.npz")
    mg = P.MultiGrid2D(ctx, [33, 33], [0, 1, 0, 1], A2, 2, np.float32)
    assert bits_equal(mg.download_v(0), g["v"]) and bits_equal(mg.download_f(0), g["f"])
    for lvl in range(1, mg.numGrids):
        assert bits_equal(mg.download_v(lvl), O.init2d([33, 33], [0, 1, 0, 1], lvl)[0])
    mg.close()
    for dtype in (np.float32, np.float64):
        n2, rg = [129, 33], [-1.5, 2, 0.25, 3]
        mg = P.MultiGrid2D(ctx, n2, rg, A2, 2, dtype)
        for lvl in range(mg.numGrids):
            assert bits_equal(mg.download_v(lvl), O.init2d(n2, rg, lvl, dtype=dtype)[0])
        mg.close()


def test_2d_mean_absolute_error_metric(ctx):
    """the thesis' accuracy metric (Fig. 4.3; PrintMeanAbsoluteError, C2/Grid2D.cu:123-154) on the device"""
    n = 129
    mg = P.MultiGrid2D(ctx, [n] * 2, [0, 20, 0, 20], A2, 2, np.float32)
    mg.FullMultiGridVCycle(0, 1, 100, 100)
    v = mg.download_v(0)
    f32 = np.float32
    hx = f32(20) / f32(n - 1)
    xs = np.array([f32(f32(0) + f32(i) * hx) for i in range(n)], f32)
    X, Y = xs[None, :], xs[:, None]
    real = (f32(2) * X * X - f32(4) * X * Y + f32(2) * Y * Y).astype(f32)
    want = np.abs((v - real)[1:-1, 1:-1].astype(np.float64)).mean()
    got = mg.MeanAbsoluteError(0)
    assert abs(got - want) <= 1e-6 * max(1.0, want)
    mg.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_3d_public_relax_on_coarse_level_with_nonzero_boundary_then_cycle(ctx, dtype):
    """public-operator use on a coarse level must not leave a stale boundary in the level's ping-pong partner: upload a v with a
    non-zero boundary into levels 1 and 2 (65^3, 33^3: the one-launch-per-sweep levels), call Relax there (the partner's boundary
    now equals that non-zero one), then run V-cycles from level 0 -- the reference zeroes the coarse v including its boundary
    (N3/MultiGrid3D.cpp:634), so the result is the oracle's plain cycle"""
    n3, rg = (129, 129, 129), [-1, 1, 0, 2, 0.5, 3]
    rng = np.random.default_rng(77)
    v = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(dtype)
    for v1 in (2, 1):  # way down: fused smooth+residual+restrict resp. relax_from_zero_pp first
        mg = P.MultiGrid3D(ctx, n3, rg, dtype)
        mg.upload_v(0, v)
        mg.upload_f(0, f)
        for lvl in (1, 2):
            m = mg.size(lvl)
            mg.upload_v(lvl, rng.uniform(1, 2, O.shape(m)).astype(dtype))
            mg.upload_f(lvl, rng.uniform(-1, 1, O.shape(m)).astype(dtype))
            mg.Relax(lvl, 2)
        mg.VCycle(0, v1, 2)
        mg.VCycle(0, v1, 2)
        want = O.cycle3d(n3, rg, mode=0, v1=v1, v2=2, reps=2, v=v, f=f, dtype=dtype)
        assert bits_equal(mg.download_v(0), want), v1
        mg.close()
