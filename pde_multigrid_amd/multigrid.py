"""ctypes views of the C host layer (include/mg_multigrid.h) and of the raw operator C-ABI
(include/mgx.h).  numpy arrays are indexed [z, y, x] (x fastest in memory): the reference's
idx = x + y*sx + z*sx*sy.  Sizes are given as (sx, sy, sz) like the reference's sizeXYZ."""
import ctypes as C

import numpy as np

from ._lib import REF_COMPAT, check, lib


def _ct(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError("dtype must be float32 or float64, got %s" % dtype)


def _ip(a):
    return (C.c_int * len(a))(*[int(x) for x in a])


def _rp(a, ct):
    return (ct * len(a))(*[float(x) for x in a])


def _shape(n):
    return tuple(int(k) for k in reversed(tuple(n)))


def _count(n):
    c = 1
    for k in n:
        c *= int(k)
    return c


def num_grids(min_size):
    return lib.mg_num_grids(int(min_size))


def coarse_size(n):
    return tuple(lib.mg_coarse_size(int(k)) for k in n)


def grid_spacing(n, rng, dtype):
    """h = range/(real)(size-1) evaluated in `dtype` like Grid3D's constructor (N3/Grid3D.cpp:31-45)."""
    t = np.dtype(dtype).type
    return [t(t(rng[2 * d + 1]) - t(rng[2 * d])) / t(int(n[d]) - 1) for d in range(len(n))]


class Context:
    """One HIP device context (compute + comm stream)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        check(lib.mgx_ctx_create(int(device), C.byref(self._h)))
        self.device = int(device)

    def close(self):
        if self._h:
            lib.mgx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(lib.mgx_ctx_sync(self._h))

    # -- raw device memory --------------------------------------------------
    def malloc(self, nbytes):
        p = C.c_void_p()
        check(lib.mgx_malloc(self._h, C.c_size_t(int(nbytes)), C.byref(p)))
        return p

    def free(self, p):
        check(lib.mgx_free(self._h, p))

    def last_relax_kernel(self):
        return lib.mgx_ctx_last_relax_kernel(self._h).decode()

    def last_rr_kernel(self):
        """the fused black pass + residual + restrict kernel of the most recent smooth_residual_restrict call ("" = not fused)"""
        return lib.mgx_ctx_last_rr_kernel(self._h).decode()

    def last_corr_kernel(self):
        """the correcting red pass of the most recent interpolate_correct_relax call ("" = the correction was a pass of its own)"""
        return lib.mgx_ctx_last_corr_kernel(self._h).decode()

    def set_param(self, name, value):
        check(lib.mgx_ctx_set_param(self._h, name.encode(), C.c_int(int(value))))

    def clear_abort(self, reenable=False):
        """after sync() reported a given-up wait between workgroups: clear the condition (mgx_ctx_clear_abort)"""
        check(lib.mgx_ctx_clear_abort(self._h, C.c_int(1 if reenable else 0)))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        check(lib.mgx_memcpy_h2d(self._h, p, arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
        return p

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype)
        check(lib.mgx_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes)))
        return out

    # -- events on the compute stream -----------------------------------------
    def event(self):
        e = C.c_void_p()
        check(lib.mgx_event_create(self._h, C.byref(e)))
        return e

    def record(self, e):
        check(lib.mgx_event_record(self._h, e))

    def elapsed_ms(self, e0, e1):
        ms = C.c_float()
        check(lib.mgx_event_elapsed_ms(self._h, e0, e1, C.byref(ms)))
        return float(ms.value)

    # -- RCCL -------------------------------------------------------------------
    @staticmethod
    def unique_id():
        buf = (C.c_ubyte * 128)()
        check(lib.mgx_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(lib.mgx_comm_init(self._h, buf, int(rank), int(nranks)))

    def comm_init_rehearsal(self, unique_id, virtual_rank, virtual_nranks):
        """one rank of a larger job rehearsed on this GPU alone (mgx_comm_init_rehearsal): timing only"""
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(lib.mgx_comm_init_rehearsal(self._h, buf, int(virtual_rank), int(virtual_nranks)))

    def comm_info(self):
        """(ranks the communicator reports, RCCL version code)"""
        n, v = C.c_int(0), C.c_int(0)
        check(lib.mgx_comm_info(self._h, C.byref(n), C.byref(v)))
        return n.value, v.value

    def comm_set_inline(self, on):
        """collectives on the compute stream (True) or on the comm stream (False): mgx_comm_set_inline"""
        check(lib.mgx_comm_set_inline(self._h, C.c_int(int(bool(on)))))


# --------------------------------------------------------------------------- raw operators
def xs_geometry(sx, itemsize):
    """(H, P): offset of the odd-x half and row pitch of the x-split layout (both multiples of a 128-byte line)"""
    al = 128 // itemsize
    H = ((sx + 1) // 2 + al - 1) // al * al
    return H, H + (sx // 2 + al - 1) // al * al


def xs_pack(a):
    """numpy restatement of the x-split layout: every x-row as its even-x half then, from the next 128-byte
    boundary on, its odd-x half; rows padded to the pitch with zeros"""
    a = np.asarray(a)
    sx = a.shape[-1]
    H, P = xs_geometry(sx, a.dtype.itemsize)
    out = np.zeros(a.shape[:-1] + (P,), a.dtype)
    out[..., :(sx + 1) // 2] = a[..., 0::2]
    out[..., H:H + sx // 2] = a[..., 1::2]
    return out


def xs_unpack(a, sx):
    a = np.asarray(a)
    H, P = xs_geometry(sx, a.dtype.itemsize)
    assert a.shape[-1] == P
    out = np.empty(a.shape[:-1] + (sx,), a.dtype)
    out[..., 0::2] = a[..., :(sx + 1) // 2]
    out[..., 1::2] = a[..., H:H + sx // 2]
    return out


class _Ops:
    """Per-operator entry points of mgx.h on numpy arrays (reference layout): upload, launch, download.
    Used by the parity tests; the cycle code in C keeps everything resident instead.  With
    xsplit=True the arrays are converted to the x-split layout on the host and the mgx3dxs_
    twins are called."""

    def __init__(self, dim, xsplit=False):
        self.dim = dim
        self.xsplit = xsplit
        self.p = ("mgx%ddxs_" if xsplit else "mgx%dd_") % dim

    def _fn(self, name, dtype):
        s, ct = _ct(dtype)
        return getattr(lib, self.p + name + "_" + s), ct

    def _run(self, ctx, arrays, call, out_index, out_shape, dtype):
        conv = xs_pack if self.xsplit else (lambda a: a)
        ptrs = [ctx.to_device(conv(np.ascontiguousarray(a, dtype=dtype))) if a is not None else None for a in arrays]
        try:
            check(call(*ptrs))
            if not self.xsplit:
                return ctx.to_host(ptrs[out_index], out_shape, dtype)
            P_ = xs_geometry(out_shape[-1], np.dtype(dtype).itemsize)[1]
            return xs_unpack(ctx.to_host(ptrs[out_index], tuple(out_shape[:-1]) + (P_,), dtype), out_shape[-1])
        finally:
            for p in ptrs:
                if p is not None:
                    ctx.free(p)

    def restrict(self, ctx, fine, n, cn=None, dtype=None):
        dtype = dtype or fine.dtype
        fn, _ = self._fn("restrict", dtype)
        cn = cn if cn is not None else coarse_size(n)
        coarse = np.zeros(_shape(cn), dtype)
        return self._run(ctx, [fine, coarse], lambda f, c: fn(ctx._h, f, _ip(n), c, _ip(cn)), 1, _shape(cn), dtype)

    def interpolate(self, ctx, fine, n, coarse, cn=None, dtype=None):
        dtype = dtype or fine.dtype
        fn, _ = self._fn("interpolate", dtype)
        cn = cn if cn is not None else coarse_size(n)
        return self._run(ctx, [fine, coarse], lambda f, c: fn(ctx._h, f, _ip(n), c, _ip(cn)), 0, _shape(n), dtype)

    def apply_correction(self, ctx, fine, n, err, en=None, dtype=None):
        dtype = dtype or fine.dtype
        fn, _ = self._fn("apply_correction", dtype)
        en = en if en is not None else n
        return self._run(ctx, [fine, err], lambda f, e: fn(ctx._h, f, _ip(n), e, _ip(en)), 0, _shape(n), dtype)

    def set(self, ctx, grid, n, value, modify_boundaries, dtype=None):
        dtype = dtype or grid.dtype
        fn, ct = self._fn("set", dtype)
        return self._run(ctx, [grid], lambda g: fn(ctx._h, g, _ip(n), ct(value), C.c_int(int(modify_boundaries))), 0,
                         _shape(n), dtype)


class _Ops3D(_Ops):
    def __init__(self, xsplit=False):
        super().__init__(3, xsplit)

    def pack(self, ctx, a):
        """device-side Natural -> XSplit conversion (mgx3dxs_pack), returned as stored (padded rows)"""
        s, _ = _ct(a.dtype)
        n = tuple(reversed(a.shape))
        fn = getattr(lib, "mgx3dxs_elems_" + s)
        fn.restype = C.c_size_t
        elems = fn(_ip(n))
        P_ = xs_geometry(n[0], a.dtype.itemsize)[1]
        assert elems == P_ * n[1] * n[2]
        src, dst = ctx.to_device(a), ctx.to_device(np.zeros(elems, a.dtype))
        try:
            check(getattr(lib, "mgx3dxs_pack_" + s)(ctx._h, src, dst, _ip(n)))
            return ctx.to_host(dst, a.shape[:-1] + (P_,), a.dtype)
        finally:
            ctx.free(src)
            ctx.free(dst)

    def unpack(self, ctx, a, sx):
        s, _ = _ct(a.dtype)
        n = (sx,) + tuple(reversed(a.shape[:-1]))
        src, dst = ctx.to_device(a), ctx.malloc(sx * a.shape[0] * a.shape[1] * a.dtype.itemsize)
        try:
            check(getattr(lib, "mgx3dxs_unpack_" + s)(ctx._h, src, dst, _ip(n)))
            return ctx.to_host(dst, a.shape[:-1] + (sx,), a.dtype)
        finally:
            ctx.free(src)
            ctx.free(dst)

    def relax(self, ctx, v, f, n, rng, ncycles, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("relax", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        return self._run(ctx, [v, f], lambda a, b: fn(ctx._h, a, b, _ip(n), h, C.c_int(ncycles)), 0, _shape(n), dtype)

    def relax_pp(self, ctx, v, f, n, rng, ncycles, w=None, w_rim_valid=False, dtype=None):
        """x-split only: relax with a ping-pong partner w (one launch per red+black sweep where the level takes it);
        w defaults to an array of NaNs (its boundary is then copied from v by the call)"""
        dtype = dtype or v.dtype
        fn, ct = self._fn("relax_pp", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        if w is None:
            w = np.full(_shape(n), np.nan, dtype)
        return self._run(ctx, [v, w, f], lambda a, ww, b: fn(ctx._h, a, ww, b, _ip(n), h, C.c_int(ncycles), C.c_int(int(w_rim_valid))),
                         0, _shape(n), dtype)

    def relax_from_zero_pp(self, ctx, v, f, n, rng, ncycles, rim_is_zero, w=None, w_rim_valid=False, dtype=None):
        """x-split only: relax_from_zero with a ping-pong partner (mgx3dxs_relax_from_zero_pp)"""
        dtype = dtype or v.dtype
        fn, ct = self._fn("relax_from_zero_pp", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        if w is None:
            w = np.full(_shape(n), np.nan, dtype)
        return self._run(ctx, [v, w, f], lambda a, ww, b: fn(ctx._h, a, ww, b, _ip(n), h, C.c_int(ncycles), C.c_int(int(rim_is_zero)),
                                                            C.c_int(int(w_rim_valid))), 0, _shape(n), dtype)

    def interpolate_correct_relax_pp(self, ctx, v, f, n, rng, coarse, ncycles, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("interpolate_correct_relax_pp", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        cn = coarse_size(n)
        w = np.full(_shape(n), np.nan, dtype)
        return self._run(ctx, [v, w, f, coarse], lambda a, ww, b, c: fn(ctx._h, a, ww, b, _ip(n), h, c, _ip(cn), C.c_int(ncycles), C.c_int(0)),
                         0, _shape(n), dtype)

    def sweep_once(self, ctx, vin, vout, f, n, rng, zero=False, dtype=None):
        """x-split only: one out-of-place sweep vin -> vout (interior of vout) by the level's one-launch kernel"""
        dtype = dtype or vin.dtype
        fn, ct = self._fn("sweep_once", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        return self._run(ctx, [vin, vout, f], lambda a, o, b: fn(ctx._h, a, o, b, _ip(n), h, C.c_int(int(zero))), 1, _shape(n), dtype)

    def relax_pp_takes(self, ctx, n, ncycles, dtype=np.float64):
        s, _ = _ct(dtype)
        return bool(getattr(lib, "mgx3dxs_relax_pp_takes_" + s)(ctx._h, _ip(n), C.c_int(ncycles)))

    def residual(self, ctx, v, f, n, rng, mode=REF_COMPAT, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("residual", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        r = np.zeros(_shape(n), dtype)
        return self._run(ctx, [v, f, r], lambda a, b, c: fn(ctx._h, a, b, c, _ip(n), h, C.c_int(mode)), 2, _shape(n), dtype)

    def residual_restrict(self, ctx, v, f, n, rng, mode=REF_COMPAT, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("residual_restrict", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        cn = coarse_size(n)
        coarse = np.zeros(_shape(cn), dtype)
        return self._run(ctx, [v, f, coarse], lambda a, b, c: fn(ctx._h, a, b, _ip(n), h, C.c_int(mode), c, _ip(cn)), 2,
                         _shape(cn), dtype)

    def interpolate_correct(self, ctx, v, n, coarse, dtype=None):
        dtype = dtype or v.dtype
        fn, _ = self._fn("interpolate_correct", dtype)
        cn = coarse_size(n)
        return self._run(ctx, [v, coarse], lambda a, c: fn(ctx._h, a, _ip(n), c, _ip(cn)), 0, _shape(n), dtype)

    def interpolate_correct_colour(self, ctx, v, n, coarse, colour, dtype=None):
        """x-split only: correct the points with (x+y+z) % 2 == colour (-1 = all)"""
        dtype = dtype or v.dtype
        fn, _ = self._fn("interpolate_correct_colour", dtype)
        cn = coarse_size(n)
        return self._run(ctx, [v, coarse], lambda a, c: fn(ctx._h, a, _ip(n), c, _ip(cn), C.c_int(colour)), 0, _shape(n), dtype)

    def relax_from_zero(self, ctx, v, f, n, rng, ncycles, rim_is_zero, dtype=None):
        """v := 0, then ncycles sweeps; with rim_is_zero the given v must have zero boundary entries (its interior is ignored)"""
        dtype = dtype or v.dtype
        fn, ct = self._fn("relax_from_zero", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        return self._run(ctx, [v, f], lambda a, b: fn(ctx._h, a, b, _ip(n), h, C.c_int(ncycles), C.c_int(int(rim_is_zero))), 0, _shape(n),
                         dtype)

    def smooth_residual_restrict(self, ctx, v, f, n, rng, ncycles, from_zero=False, v_rim_is_zero=False, mode=REF_COMPAT, dtype=None):
        """x-split only: (v_out, coarse_f) of Relax(ncycles) + CalculateResidual + Restrict in one call
        (mgx3dxs_smooth_residual_restrict: the last black pass inside the residual+restrict launch where the level takes it)"""
        dtype = dtype or v.dtype
        fn, ct = self._fn("smooth_residual_restrict", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        cn = coarse_size(n)
        pv, pf = ctx.to_device(xs_pack(np.ascontiguousarray(v, dtype))), ctx.to_device(xs_pack(np.ascontiguousarray(f, dtype)))
        pc = ctx.to_device(xs_pack(np.full(_shape(cn), np.nan, dtype)))  # the call has to zero the coarse boundary itself
        try:
            check(fn(ctx._h, pv, pf, _ip(n), h, C.c_int(ncycles), C.c_int(int(from_zero)), C.c_int(int(v_rim_is_zero)), C.c_int(mode), pc,
                     _ip(cn), C.c_int(0)))
            isz = np.dtype(dtype).itemsize
            out_v = xs_unpack(ctx.to_host(pv, tuple(_shape(n)[:-1]) + (xs_geometry(n[0], isz)[1],), dtype), n[0])
            out_c = xs_unpack(ctx.to_host(pc, tuple(_shape(cn)[:-1]) + (xs_geometry(cn[0], isz)[1],), dtype), cn[0])
            return out_v, out_c
        finally:
            for q in (pv, pf, pc):
                ctx.free(q)

    def interpolate_correct_relax(self, ctx, v, f, n, rng, coarse, ncycles, dtype=None):
        """x-split only: v += Interpolate(coarse) on the interior, then ncycles >= 1 red-black sweeps, in one call"""
        dtype = dtype or v.dtype
        fn, ct = self._fn("interpolate_correct_relax", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        cn = coarse_size(n)
        return self._run(ctx, [v, f, coarse], lambda a, b, c: fn(ctx._h, a, b, _ip(n), h, c, _ip(cn), C.c_int(ncycles)), 0, _shape(n),
                         dtype)

    def jacobi(self, ctx, v, f, n, rng, omega, ncycles, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("jacobi", dtype)
        h = _rp(grid_spacing(n, rng, dtype), ct)
        return self._run(ctx, [v, np.zeros_like(v), f], lambda a, t, b: fn(ctx._h, a, t, b, _ip(n), h, ct(omega), C.c_int(ncycles)),
                         0, _shape(n), dtype)

    def norm2(self, ctx, x):
        s, _ = _ct(x.dtype)
        out = C.c_double()
        p = ctx.to_device(x)
        try:
            check(getattr(lib, "mgx_norm2_" + s)(ctx._h, p, C.c_size_t(x.size), C.byref(out)))
        finally:
            ctx.free(p)
        return float(out.value)


class _Ops2D(_Ops):
    def __init__(self):
        super().__init__(2)

    def _geom(self, n, rng, A, dtype, ct):
        t = np.dtype(dtype).type
        return (_rp(grid_spacing(n, rng, dtype), ct), _rp([t(rng[0]), t(rng[2])], ct), _rp(A, ct))

    def relax(self, ctx, v, f, n, rng, A, alfa, ncycles, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("relax", dtype)
        h, a, AA = self._geom(n, rng, A, dtype, ct)
        return self._run(ctx, [v, f], lambda x, y: fn(ctx._h, x, y, _ip(n), h, a, AA, C.c_int(alfa), C.c_int(ncycles)), 0,
                         _shape(n), dtype)

    def jacobi(self, ctx, v, f, n, rng, A, alfa, omega, ncycles, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("jacobi", dtype)
        h, a, AA = self._geom(n, rng, A, dtype, ct)
        return self._run(ctx, [v, np.zeros_like(v), f],
                         lambda x, t, y: fn(ctx._h, x, t, y, _ip(n), h, a, AA, C.c_int(alfa), ct(omega), C.c_int(ncycles)), 0,
                         _shape(n), dtype)

    def residual(self, ctx, v, f, n, rng, A, alfa, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("residual", dtype)
        h, a, AA = self._geom(n, rng, A, dtype, ct)
        r = np.zeros(_shape(n), dtype)
        return self._run(ctx, [v, f, r], lambda x, y, z: fn(ctx._h, x, y, z, _ip(n), h, a, AA, C.c_int(alfa)), 2, _shape(n),
                         dtype)


    def residual_restrict(self, ctx, v, f, n, rng, A, alfa, dtype=None):
        dtype = dtype or v.dtype
        fn, ct = self._fn("residual_restrict", dtype)
        h, a, AA = self._geom(n, rng, A, dtype, ct)
        cn = coarse_size(n)
        c = np.zeros(_shape(cn), dtype)
        return self._run(ctx, [v, f, c], lambda x, y, z: fn(ctx._h, x, y, _ip(n), h, a, AA, C.c_int(alfa), z, _ip(cn)), 2,
                         _shape(cn), dtype)

    def interpolate_correct(self, ctx, v, n, coarse, dtype=None):
        dtype = dtype or v.dtype
        fn, _ = self._fn("interpolate_correct", dtype)
        cn = coarse_size(n)
        return self._run(ctx, [v, coarse], lambda x, c: fn(ctx._h, x, _ip(n), c, _ip(cn)), 0, _shape(n), dtype)


def _ops2d_fused_down(self, ctx, v, f, n, rng, A, alfa, ncycles, v_zero=False, restrict=True, dtype=None):
    """(v_out, coarse_f) of mgx2d_relax_residual_restrict"""
    dtype = dtype or v.dtype
    fn, ct = self._fn("relax_residual_restrict", dtype)
    h, a, AA = self._geom(n, rng, A, dtype, ct)
    cn = coarse_size(n)
    pv, pf = ctx.to_device(np.ascontiguousarray(v, dtype)), ctx.to_device(np.ascontiguousarray(f, dtype))
    po, pc = ctx.to_device(np.full(_shape(n), np.nan, dtype)), ctx.to_device(np.full(_shape(cn), np.nan, dtype))
    try:
        check(fn(ctx._h, pv, po, pf, _ip(n), h, a, AA, C.c_int(alfa), C.c_int(ncycles), C.c_int(int(v_zero)),
                 pc if restrict else None, _ip(cn) if restrict else None))
        return ctx.to_host(po, _shape(n), dtype), (ctx.to_host(pc, _shape(cn), dtype) if restrict else None)
    finally:
        for p in (pv, pf, po, pc):
            ctx.free(p)


def _ops2d_fused_up(self, ctx, v, f, n, rng, A, alfa, coarse, ncycles, dtype=None):
    dtype = dtype or v.dtype
    fn, ct = self._fn("interpolate_correct_relax", dtype)
    h, a, AA = self._geom(n, rng, A, dtype, ct)
    cn = coarse_size(n)
    pv, pf = ctx.to_device(np.ascontiguousarray(v, dtype)), ctx.to_device(np.ascontiguousarray(f, dtype))
    po, pc = ctx.to_device(np.full(_shape(n), np.nan, dtype)), ctx.to_device(np.ascontiguousarray(coarse, dtype))
    try:
        check(fn(ctx._h, pv, po, pf, _ip(n), h, a, AA, C.c_int(alfa), pc, _ip(cn), C.c_int(ncycles)))
        return ctx.to_host(po, _shape(n), dtype)
    finally:
        for p in (pv, pf, po, pc):
            ctx.free(p)


_Ops2D.relax_residual_restrict = _ops2d_fused_down
_Ops2D.interpolate_correct_relax = _ops2d_fused_up

ops3d = _Ops3D()
ops3dxs = _Ops3D(xsplit=True)
ops2d = _Ops2D()


# --------------------------------------------------------------------------- hierarchy views
def _grid3_struct(ct):
    class Grid3D(C.Structure):
        _fields_ = [("h_v", C.c_void_p), ("h_f", C.c_void_p), ("d_v", C.c_void_p), ("d_f", C.c_void_p),
                    ("d_r", C.c_void_p), ("d_e", C.c_void_p), ("sizeX", C.c_int), ("sizeY", C.c_int), ("sizeZ", C.c_int),
                    ("sizeXYZ", C.c_int * 3), ("h_x", ct), ("h_y", ct), ("h_z", ct), ("x_a", ct), ("x_b", ct),
                    ("y_a", ct), ("y_b", ct), ("z_a", ct), ("z_b", ct)]

    class MultiGrid3D(C.Structure):
        _fields_ = [("grids3D", C.POINTER(C.POINTER(Grid3D))), ("numGrids", C.c_int), ("maxGrids", C.c_int),
                    ("ctx", C.c_void_p), ("residual_mode", C.c_int), ("fuse", C.c_int), ("layout", C.c_int),
                    ("smoother", C.c_int), ("omega", ct), ("use_graph", C.c_int), ("capturing", C.c_int),
                    ("graph_exec", C.c_void_p * 32), ("graph_key", C.c_longlong * 32), ("f_rim_zero", C.c_ubyte * 32),
                    ("v_rim_zero", C.c_ubyte * 32), ("e_rim_valid", C.c_ubyte * 32)]

    return Grid3D, MultiGrid3D


def _grid2_struct(ct):
    class Grid2D(C.Structure):
        _fields_ = [("h_v", C.c_void_p), ("h_f", C.c_void_p), ("d_v", C.c_void_p), ("d_f", C.c_void_p),
                    ("d_r", C.c_void_p), ("d_e", C.c_void_p), ("sizeX", C.c_int), ("sizeY", C.c_int),
                    ("sizeXY", C.c_int * 2), ("h_x", ct), ("h_y", ct), ("x_a", ct), ("x_b", ct), ("y_a", ct), ("y_b", ct)]

    class MultiGrid2D(C.Structure):
        _fields_ = [("grids2D", C.POINTER(C.POINTER(Grid2D))), ("numGrids", C.c_int), ("maxGrids", C.c_int),
                    ("matrixA", ct * 4), ("sizeA", C.c_int), ("alfa", C.c_int), ("ctx", C.c_void_p), ("fuse", C.c_int),
                    ("smoother", C.c_int), ("omega", ct), ("use_graph", C.c_int), ("capturing", C.c_int),
                    ("graph_exec", C.c_void_p * 32), ("graph_key", C.c_longlong * 32)]

    return Grid2D, MultiGrid2D


def _grid1_struct(ct):
    class Grid1D(C.Structure):
        _fields_ = [("h_v", C.POINTER(ct)), ("h_f", C.POINTER(ct)), ("sizeX", C.c_int), ("h_x", ct), ("x_a", ct),
                    ("x_b", ct)]

    class MultiGrid1D(C.Structure):
        _fields_ = [("grids1D", C.POINTER(C.POINTER(Grid1D))), ("numGrids", C.c_int), ("maxGrids", C.c_int)]

    return Grid1D, MultiGrid1D


class _MGBase:
    _prefix = None

    def _call(self, name, *args):
        return check(getattr(lib, "%s_%s_%s" % (self._prefix, self._sfx, name))(self._mg, *args))

    @property
    def numGrids(self):
        return self._mg.contents.numGrids

    @numGrids.setter
    def numGrids(self, k):
        if not 1 <= int(k) <= self._mg.contents.maxGrids:
            raise ValueError("numGrids must be in [1, %d]" % self._mg.contents.maxGrids)
        self._mg.contents.numGrids = int(k)

    @property
    def use_graph(self):
        """2D / 3D: capture VCycle into a HIP graph on first use and replay it (mg_multigrid.h)"""
        return bool(self._mg.contents.use_graph)

    @use_graph.setter
    def use_graph(self, on):
        self._mg.contents.use_graph = 1 if on else 0

    @property
    def maxGrids(self):
        return self._mg.contents.maxGrids

    def set_smoother(self, name, omega=None):
        """'rbgs' = red-black Gauss-Seidel (the reference's smoother), 'jacobi' = weighted Jacobi (addition)"""
        self._mg.contents.smoother = {"rbgs": 0, "jacobi": 1}[name]
        if omega is not None:
            self._mg.contents.omega = float(omega)

    def VCycle(self, gridID, v1, v2):
        self._call("VCycle", C.c_int(gridID), C.c_int(v1), C.c_int(v2))

    def FullMultiGridVCycle(self, gridID, v0, v1, v2):
        self._call("FullMultiGridVCycle", C.c_int(gridID), C.c_int(v0), C.c_int(v1), C.c_int(v2))

    def close(self):
        if self._mg:
            getattr(lib, "%s_%s_destroy" % (self._prefix, self._sfx))(self._mg)
            self._mg = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGrid3D(_MGBase):
    """MultiGrid3D(finestGridSizeXYZ, range) of the reference (N3/MultiGrid3D.h:6-33) on one MI355X."""
    _prefix = "mgMultiGrid3D"

    def __init__(self, ctx, finestGridSizeXYZ, rng, dtype=np.float64, nlevels=0, residual_mode=REF_COMPAT, fuse=True,
                 layout="xsplit"):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self._sfx, self._ct = _ct(dtype)
        self._G, self._M = _grid3_struct(self._ct)
        self._mg = C.POINTER(self._M)()
        fn = getattr(lib, "mgMultiGrid3D_%s_create_layout" % self._sfx)
        lay = {"natural": 0, "xsplit": 1}[layout]
        check(fn(ctx._h, _ip(finestGridSizeXYZ), _rp(rng, self._ct), C.c_int(lay), C.byref(self._mg)))
        if nlevels:
            self.numGrids = nlevels
        self._mg.contents.residual_mode = int(residual_mode)
        self._mg.contents.fuse = int(bool(fuse))

    def grid(self, gridID):
        return self._mg.contents.grids3D[gridID].contents

    def size(self, gridID):
        return tuple(self.grid(gridID).sizeXYZ)

    def Relax(self, gridID, ncycles):
        self._call("Relax", self._mg.contents.grids3D[gridID], C.c_int(ncycles))

    def CalculateResidual(self, gridID):
        out = np.empty(_shape(self.size(gridID)), self.dtype)
        self._call("download_residual", C.c_int(gridID), out.ctypes.data_as(C.c_void_p))
        return out

    def ResidualNorm(self, gridID=0):
        out = C.c_double()
        self._call("ResidualNorm", C.c_int(gridID), C.byref(out))
        return float(out.value)

    def InitF(self, gridID):
        self._call("InitF", C.c_int(gridID))

    def DiffStats(self, gridID=0):
        """(mean |diff|, max |diff|, relative L2) of diff = analytic - v  (Grid3D::PrintDiff as numbers)"""
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._call("DiffStats", C.c_int(gridID), C.byref(a), C.byref(b), C.byref(c))
        return float(a.value), float(b.value), float(c.value)

    def setToValue_v(self, gridID, value, modifyBoundaries):
        g = self.grid(gridID)
        self._call("setToValue", C.c_void_p(g.d_v), _ip(g.sizeXYZ), self._ct(value), C.c_int(int(modifyBoundaries)))

    def interpolate_correct(self, gridID):
        """v[gridID] += Interpolate(v[gridID+1]) on the interior (the fused VCycle step)."""
        g0, g1 = self.grid(gridID), self.grid(gridID + 1)
        pfx = "mgx3dxs_" if self._mg.contents.layout else "mgx3d_"
        check(getattr(lib, pfx + "interpolate_correct_" + self._sfx)(self.ctx._h, C.c_void_p(g0.d_v), _ip(g0.sizeXYZ),
                                                                     C.c_void_p(g1.d_v), _ip(g1.sizeXYZ)))

    def upload_v(self, gridID, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        assert arr.size == _count(self.size(gridID))
        self._call("upload_v", C.c_int(gridID), arr.ctypes.data_as(C.c_void_p))

    def upload_f(self, gridID, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        assert arr.size == _count(self.size(gridID))
        self._call("upload_f", C.c_int(gridID), arr.ctypes.data_as(C.c_void_p))

    def download_v(self, gridID=0):
        out = np.empty(_shape(self.size(gridID)), self.dtype)
        self._call("download_v", C.c_int(gridID), out.ctypes.data_as(C.c_void_p))
        return out

    def download_f(self, gridID=0):
        out = np.empty(_shape(self.size(gridID)), self.dtype)
        self._call("download_f", C.c_int(gridID), out.ctypes.data_as(C.c_void_p))
        return out


class MultiGrid2D(_MGBase):
    """MultiGrid2D(finestGridSizeXY, range, A, A_size, alfa) (N2/MultiGrid2D.h:6-37) on one MI355X."""
    _prefix = "mgMultiGrid2D"

    def __init__(self, ctx, finestGridSizeXY, rng, A, alfa, dtype=np.float64, nlevels=0, fuse=True):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self._sfx, self._ct = _ct(dtype)
        self._G, self._M = _grid2_struct(self._ct)
        self._mg = C.POINTER(self._M)()
        fn = getattr(lib, "mgMultiGrid2D_%s_create" % self._sfx)
        check(fn(ctx._h, _ip(finestGridSizeXY), _rp(rng, self._ct), _rp(A, self._ct), C.c_int(2), C.c_int(alfa),
                 C.byref(self._mg)))
        # fuse: True = the library default (2: cache-resident cycle kernels), 1 = fused operators only, False / 0 = one
        # launch per reference call
        self._mg.contents.fuse = 2 if fuse is True else int(fuse)
        if nlevels:
            self.numGrids = nlevels

    def grid(self, gridID):
        return self._mg.contents.grids2D[gridID].contents

    def MeanAbsoluteError(self, gridID=0):
        out = C.c_double()
        self._call("MeanAbsoluteError", C.c_int(gridID), C.byref(out))
        return float(out.value)

    def size(self, gridID):
        return tuple(self.grid(gridID).sizeXY)

    def Relax(self, gridID, ncycles):
        self._call("Relax", self._mg.contents.grids2D[gridID], C.c_int(ncycles))

    def upload_v(self, gridID, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        assert arr.size == _count(self.size(gridID))
        self._call("upload_v", C.c_int(gridID), arr.ctypes.data_as(C.c_void_p))

    def upload_f(self, gridID, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        assert arr.size == _count(self.size(gridID))
        self._call("upload_f", C.c_int(gridID), arr.ctypes.data_as(C.c_void_p))

    def download_v(self, gridID=0):
        out = np.empty(_shape(self.size(gridID)), self.dtype)
        self._call("download_v", C.c_int(gridID), out.ctypes.data_as(C.c_void_p))
        return out

    def download_f(self, gridID=0):
        out = np.empty(_shape(self.size(gridID)), self.dtype)
        self._call("download_f", C.c_int(gridID), out.ctypes.data_as(C.c_void_p))
        return out


class MultiGrid1D(_MGBase):
    """MultiGrid1D(finestGridSize, range) (N1/MultiGrid1D.h:6-31): host-only C (BASELINE configs[0])."""
    _prefix = "mgMultiGrid1D"

    def __init__(self, finestGridSize, rng, dtype=np.float32, nlevels=0):
        self.dtype = np.dtype(dtype)
        self._sfx, self._ct = _ct(dtype)
        self._G, self._M = _grid1_struct(self._ct)
        self._mg = C.POINTER(self._M)()
        fn = getattr(lib, "mgMultiGrid1D_%s_create" % self._sfx)
        check(fn(C.c_int(finestGridSize), _rp(rng, self._ct), C.byref(self._mg)))
        if nlevels:
            self.numGrids = nlevels

    def grid(self, gridID):
        return self._mg.contents.grids1D[gridID].contents

    def _view(self, ptr, n):
        return np.ctypeslib.as_array(ptr, shape=(n,))

    def v(self, gridID=0):
        g = self.grid(gridID)
        return self._view(g.h_v, g.sizeX)

    def f(self, gridID=0):
        g = self.grid(gridID)
        return self._view(g.h_f, g.sizeX)

    def Relax(self, gridID, ncycles):
        self._call("Relax", self._mg.contents.grids1D[gridID], C.c_int(ncycles))

    def CalculateResidual(self, gridID):
        g = self.grid(gridID)
        r = np.empty(g.sizeX, self.dtype)
        self._call("CalculateResidual", self._mg.contents.grids1D[gridID], r.ctypes.data_as(C.c_void_p))
        return r

    def Restrict(self, fine):
        fine = np.ascontiguousarray(fine, self.dtype)
        coarse = np.empty((fine.size - 1) // 2 + 1, self.dtype)
        self._call("Restrict", fine.ctypes.data_as(C.c_void_p), C.c_int(fine.size), coarse.ctypes.data_as(C.c_void_p),
                   C.c_int(coarse.size))
        return coarse

    def Interpolate(self, fine, coarse):
        fine = np.ascontiguousarray(fine, self.dtype).copy()
        coarse = np.ascontiguousarray(coarse, self.dtype)
        self._call("Interpolate", fine.ctypes.data_as(C.c_void_p), C.c_int(fine.size), coarse.ctypes.data_as(C.c_void_p),
                   C.c_int(coarse.size))
        return fine

    def ApplyCorrection(self, fine, err):
        fine = np.ascontiguousarray(fine, self.dtype).copy()
        err = np.ascontiguousarray(err, self.dtype)
        self._call("ApplyCorrection", fine.ctypes.data_as(C.c_void_p), C.c_int(fine.size), err.ctypes.data_as(C.c_void_p),
                   C.c_int(err.size))
        return fine

    def setToValue(self, grid, value, modifyBoundaries):
        grid = np.ascontiguousarray(grid, self.dtype).copy()
        self._call("setToValue", grid.ctypes.data_as(C.c_void_p), C.c_int(grid.size), self._ct(value),
                   C.c_int(int(modifyBoundaries)))
        return grid


# --------------------------------------------------------------------------- z-slab decomposition
class SlabPlan(C.Structure):
    _fields_ = [("zlo", C.c_int), ("zhi", C.c_int), ("glo", C.c_int), ("ghi", C.c_int), ("zoff", C.c_int), ("nzl", C.c_int),
                ("ubeg", C.c_int), ("uend", C.c_int)]


def dist_num_levels(sizeZ, nranks, numGrids, min_planes=4):
    return lib.mg_dist_num_levels(int(sizeZ), int(nranks), int(numGrids), int(min_planes))


def slab_plan(sizeZ, rank, nranks):
    p = SlabPlan()
    check(lib.mg_slab_plan(int(sizeZ), int(rank), int(nranks), C.byref(p)))
    return p


class LocalGroup:
    """In-process test transport: `nranks` host threads, one Context each, same device (mgx_comm_init_local)."""

    def __init__(self, nranks):
        self._g = C.c_void_p()
        check(lib.mgx_local_group_create(int(nranks), C.byref(self._g)))
        self.nranks = nranks

    def attach(self, ctx, rank):
        check(lib.mgx_comm_init_local(ctx._h, self._g, int(rank)))

    def set_test_hooks(self, delay_us=0, drop_waits=False):
        """delay_us: every transfer starts that late on the receiving comm stream; drop_waits: mgx_comm_wait becomes a
        no-op (fault injection -- results must then be WRONG, which is what the negative test checks)"""
        check(lib.mgx_local_group_set_test_hooks(self._g, int(delay_us), int(bool(drop_waits))))

    def close(self):
        if self._g:
            lib.mgx_local_group_destroy(self._g)
            self._g = C.c_void_p()


def _dist_struct(ct):
    class Slab3D(C.Structure):
        _fields_ = [("d_v", C.c_void_p), ("d_f", C.c_void_p), ("sizeXYZ", C.c_int * 3), ("plan", SlabPlan), ("h_x", ct),
                    ("h_y", ct), ("h_z", ct), ("x_a", ct), ("y_a", ct), ("z_a", ct), ("level", C.c_int)]

    class DistMultiGrid3D(C.Structure):
        _fields_ = [("slabs", C.POINTER(C.POINTER(Slab3D))), ("numDist", C.c_int), ("numGrids", C.c_int), ("maxGrids", C.c_int),
                    ("tail", C.c_void_p), ("ctx", C.c_void_p), ("rank", C.c_int), ("nranks", C.c_int),
                    ("residual_mode", C.c_int), ("d_share", C.c_void_p), ("d_bplane", C.c_void_p), ("d_norm", C.c_void_p),
                    ("norm_count", C.c_int), ("inline_bytes", C.c_longlong), ("v_rim_zero", C.c_ubyte * 32),
                    ("use_graph", C.c_int), ("graph_exec", C.c_void_p), ("graph_key", C.c_longlong), ("graph_warm", C.c_int),
                    ("pack_halos", C.c_int), ("d_stage", C.c_void_p), ("stage_half", C.c_size_t),
                    ("ca_min_planes", C.c_int), ("gv", C.c_byte * 32), ("gf", C.c_byte * 32), ("comm_pending", C.c_int),
                    ("n_exchanges", C.c_longlong)]

    return Slab3D, DistMultiGrid3D


class DistMultiGrid3D(_MGBase):
    """One rank's part of a z-slab decomposed 3D hierarchy (include/mg_multigrid.h, mgDistMultiGrid3D_<r>).
    The context must already carry a communicator (Context.comm_init for RCCL, LocalGroup.attach for the
    in-process test transport) unless it runs alone."""
    _prefix = "mgDistMultiGrid3D"

    def __init__(self, ctx, finestGridSizeXYZ, rng, dtype=np.float64, nlevels=0, residual_mode=REF_COMPAT, min_planes=4,
                 inline_bytes=None, use_graph=False, pack_halos=None, ca_min_planes=None):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self._sfx, self._ct = _ct(dtype)
        self._S, self._M = _dist_struct(self._ct)
        self._mg = C.POINTER(self._M)()
        self.n = tuple(int(k) for k in finestGridSizeXYZ)
        fn = getattr(lib, "mgDistMultiGrid3D_%s_create" % self._sfx)
        check(fn(ctx._h, _ip(finestGridSizeXYZ), _rp(rng, self._ct), C.c_int(min_planes), C.byref(self._mg)))
        if nlevels:
            self.numGrids = nlevels
        self._mg.contents.residual_mode = int(residual_mode)
        if inline_bytes is not None:  # None: the library default (mg_multigrid.h); 0: every level overlapped
            self._mg.contents.inline_bytes = int(inline_bytes)
        self._mg.contents.use_graph = int(bool(use_graph))  # opt-in: VCycle(0, ...) captured (RCCL calls included) and replayed
        if pack_halos is not None:  # None: the library default (0 = whole planes)
            self._mg.contents.pack_halos = int(bool(pack_halos))
        if ca_min_planes is not None:  # None: the library default (16); 0: one exchange per colour pass on every level
            self._mg.contents.ca_min_planes = int(ca_min_planes)

    @property
    def n_exchanges(self):
        """halo exchanges + collectives this rank has enqueued since the hierarchy was created"""
        return int(self._mg.contents.n_exchanges)

    @property
    def inline_bytes(self):
        return self._mg.contents.inline_bytes

    @property
    def numDist(self):
        return self._mg.contents.numDist

    @property
    def rank(self):
        return self._mg.contents.rank

    @property
    def nranks(self):
        return self._mg.contents.nranks

    def plan(self, gridID=0):
        return self._mg.contents.slabs[gridID].contents.plan

    def Relax(self, gridID, ncycles):
        self._call("Relax", C.c_int(gridID), C.c_int(ncycles))

    def zero_v(self, gridID=0):
        self._call("zero_v", C.c_int(gridID))

    def ResidualNorm(self, gridID=0):
        """l2 norm of the residual over the whole grid (slab sums + all-reduce); the same value on every rank"""
        out = C.c_double()
        self._call("ResidualNorm", C.c_int(gridID), C.byref(out))
        return float(out.value)

    def ResidualNormRecord(self, gridID=0):
        self._call("ResidualNormRecord", C.c_int(gridID))

    def ResidualNormHistory(self):
        buf = (C.c_double * 255)()
        n = C.c_int()
        self._call("ResidualNormHistory", buf, C.c_int(255), C.byref(n))
        return [float(buf[i]) for i in range(n.value)]

    def slab(self, gridID=0):
        return self._mg.contents.slabs[gridID].contents

    def relax_colour_local(self, gridID, colour):
        """one colour pass on this rank's slab WITHOUT the ghost exchange (kernel timing only)"""
        g = self.slab(gridID)
        p = g.plan
        h = (self._ct * 3)(g.h_x, g.h_y, g.h_z)
        check(getattr(lib, "mgx3dxs_relax_colour_slab_" + self._sfx)(self.ctx._h, C.c_void_p(g.d_v), C.c_void_p(g.d_f),
                                                                      C.c_int(g.sizeXYZ[0]), C.c_int(g.sizeXYZ[1]), h,
                                                                      C.c_int(colour), C.c_int(p.ubeg - p.zoff),
                                                                      C.c_int(p.uend - p.zoff), C.c_int(p.zoff)))

    def upload_v(self, gridID, full):
        full = np.ascontiguousarray(full, self.dtype)
        self._call("upload_v", C.c_int(gridID), full.ctypes.data_as(C.c_void_p))

    def upload_f(self, gridID, full):
        full = np.ascontiguousarray(full, self.dtype)
        self._call("upload_f", C.c_int(gridID), full.ctypes.data_as(C.c_void_p))

    def download_owned(self, gridID=0):
        """this rank's owned planes [plan.zlo, plan.zhi) of level gridID as an array of its own (reference layout)"""
        p = self.plan(gridID)
        g = self.slab(gridID)
        sx, sy = g.sizeXYZ[0], g.sizeXYZ[1]
        out = np.empty((p.zhi - p.zlo, sy, sx), self.dtype)
        # the C entry addresses planes of a whole-grid host array: hand it the address plane 0 would have
        base = out.ctypes.data - p.zlo * sx * sy * self.dtype.itemsize
        self._call("download_v", C.c_int(gridID), C.c_void_p(base))
        return out

    def download_v_into(self, gridID, full):
        """writes this rank's owned planes into `full` (a whole-grid array in the reference layout)"""
        assert full.dtype == self.dtype and full.flags.c_contiguous
        self._call("download_v", C.c_int(gridID), full.ctypes.data_as(C.c_void_p))


# --------------------------------------------------------------------------- solve(grid, rhs, nlevels)
def solve3d(ctx, grid, rhs, rng, nlevels=0, fmg=False, v0=1, v1=2, v2=2, ncycles=1, residual_mode=REF_COMPAT):
    grid = np.ascontiguousarray(grid).copy()
    s, ct = _ct(grid.dtype)
    rhs = np.ascontiguousarray(rhs, grid.dtype)
    n = tuple(reversed(grid.shape))
    check(getattr(lib, "mg3d_solve_" + s)(ctx._h, grid.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p), _ip(n),
                                          _rp(rng, ct), C.c_int(nlevels), C.c_int(int(fmg)), C.c_int(v0), C.c_int(v1),
                                          C.c_int(v2), C.c_int(ncycles), C.c_int(residual_mode)))
    return grid


def solve3d_from_zero(ctx, n, rng, dtype=np.float64, rhs=None, nlevels=0, fmg=False, v0=1, v1=2, v2=2, ncycles=1,
                      residual_mode=REF_COMPAT):
    """mg3d_solve_from_zero: the guess is the reference's InitV state (zeros, nothing uploaded); rhs=None: the reference's
    own right-hand side, built on the device -- then the only transfer of the call is the result"""
    s, ct = _ct(dtype)
    out = np.empty(_shape(n), dtype)
    r = np.ascontiguousarray(rhs, dtype).ctypes.data_as(C.c_void_p) if rhs is not None else None
    check(getattr(lib, "mg3d_solve_from_zero_" + s)(ctx._h, out.ctypes.data_as(C.c_void_p), r, _ip(n), _rp(rng, ct), C.c_int(nlevels),
                                                    C.c_int(int(fmg)), C.c_int(v0), C.c_int(v1), C.c_int(v2), C.c_int(ncycles),
                                                    C.c_int(residual_mode)))
    return out


def solve2d(ctx, grid, rhs, rng, A, alfa, nlevels=0, fmg=False, v0=1, v1=2, v2=2, ncycles=1):
    grid = np.ascontiguousarray(grid).copy()
    s, ct = _ct(grid.dtype)
    rhs = np.ascontiguousarray(rhs, grid.dtype)
    n = tuple(reversed(grid.shape))
    check(getattr(lib, "mg2d_solve_" + s)(ctx._h, grid.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p), _ip(n),
                                          _rp(rng, ct), _rp(A, ct), C.c_int(alfa), C.c_int(nlevels), C.c_int(int(fmg)),
                                          C.c_int(v0), C.c_int(v1), C.c_int(v2), C.c_int(ncycles)))
    return grid


def solve1d(grid, rhs, rng, nlevels=0, fmg=False, v0=1, v1=2, v2=2, ncycles=1):
    grid = np.ascontiguousarray(grid).copy()
    s, ct = _ct(grid.dtype)
    rhs = np.ascontiguousarray(rhs, grid.dtype)
    check(getattr(lib, "mg1d_solve_" + s)(grid.ctypes.data_as(C.c_void_p), rhs.ctypes.data_as(C.c_void_p),
                                          C.c_int(grid.size), _rp(rng, ct), C.c_int(nlevels), C.c_int(int(fmg)),
                                          C.c_int(v0), C.c_int(v1), C.c_int(v2), C.c_int(ncycles)))
    return grid
