"""GPU suite: the give-up path of the kernels whose workgroups wait for each other (csrc/mgx_sync.hpp: resident Relax,
mgx_resident3d.hip; one-launch red+black sweep of 513-point rows, mgx_sweep3d.hip).  A wait that never ends must (1) end:
every wave terminates within "sync.spin_limit" polls, (2) be seen by the host: mgx_ctx_sync reports it, (3) not poison the
context: after mgx_ctx_clear_abort the same context computes bit-exact results again -- by colour passes, and with the kernels
re-enabled -- and a fresh context is unaffected.  The never-ending wait is made by the test hook "test.handoff_fault":
workgroup 0 waits for tags of a launch epoch nobody writes.  Reference operator: MultiGrid3D::Relax, N3/MultiGrid3D.cpp:489-567."""
import threading
import time

import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal

pytestmark = pytest.mark.gpu
RG = [-1, 1, 0, 2, 0.5, 3]


def _data(n3, dtype, seed=0):
    r = np.random.default_rng(seed)
    shape = tuple(reversed(n3))
    return r.uniform(-1, 1, shape).astype(dtype), r.uniform(-1, 1, shape).astype(dtype)


@pytest.mark.parametrize("form", [1, 2], ids=["per_sweep", "per_pass"])
def test_resident_wait_gives_up_is_reported_and_clears(form):
    n3, ncycles, dtype = (65, 65, 65), 6, np.float64
    v, f = _data(n3, dtype, seed=form)
    want = O.relax3d(n3, RG, v, f, ncycles, dtype=dtype)
    ctx = P.Context(0)
    try:
        ctx.set_param("relax3d.resident", form)
        ctx.set_param("sync.spin_limit", 5000)
        ctx.set_param("test.handoff_fault", 7)
        t0 = time.time()
        P.ops3dxs.relax(ctx, v, f, n3, RG, ncycles)
        assert ctx.last_relax_kernel().startswith("relax3d_xs_resident"), ctx.last_relax_kernel()
        with pytest.raises(P.MgxError, match="gave up"):
            ctx.sync()
        assert time.time() - t0 < 5.0, "the launch did not terminate within the spin limit"
        with pytest.raises(P.MgxError, match="gave up"):  # sticky until cleared
            ctx.sync()
        ctx.set_param("test.handoff_fault", 0)
        # cleared, kernels NOT re-enabled: the same call runs colour passes, bit-exact
        ctx.clear_abort(False)
        ctx.sync()
        got = P.ops3dxs.relax(ctx, v, f, n3, RG, ncycles)
        assert not ctx.last_relax_kernel().startswith("relax3d_xs_resident"), ctx.last_relax_kernel()
        ctx.sync()
        assert bits_equal(got, want)
        # re-enabled: the resident kernel again, bit-exact, several launches (epoch logic after the reset)
        ctx.clear_abort(True)
        for _ in range(3):
            got = P.ops3dxs.relax(ctx, v, f, n3, RG, ncycles)
            assert ctx.last_relax_kernel().startswith("relax3d_xs_resident"), ctx.last_relax_kernel()
            ctx.sync()
            assert bits_equal(got, want)
    finally:
        ctx.close()
    # a fresh context is unaffected
    ctx = P.Context(0)
    try:
        got = P.ops3dxs.relax(ctx, v, f, n3, RG, ncycles)
        assert ctx.last_relax_kernel().startswith("relax3d_xs_resident"), ctx.last_relax_kernel()
        ctx.sync()
        assert bits_equal(got, want)
    finally:
        ctx.close()


def test_abort_without_check_still_falls_back_after_sync_error():
    """a caller that ignores the error of sync(): every later call on the context runs colour passes (right results) and every
    later sync keeps failing until clear_abort"""
    n3, dtype = (33, 33, 33), np.float32
    v, f = _data(n3, dtype, seed=3)
    ctx = P.Context(0)
    try:
        ctx.set_param("sync.spin_limit", 3000)
        ctx.set_param("test.handoff_fault", 1)
        P.ops3dxs.relax(ctx, v, f, n3, RG, 4)
        with pytest.raises(P.MgxError):
            ctx.sync()
        ctx.set_param("test.handoff_fault", 0)
        got = P.ops3dxs.relax(ctx, v, f, n3, RG, 4)  # ops3dxs.relax downloads without checking
        assert not ctx.last_relax_kernel().startswith("relax3d_xs_resident")
        assert bits_equal(got, O.relax3d(n3, RG, v, f, 4, dtype=dtype))
        with pytest.raises(P.MgxError):
            ctx.sync()
        ctx.clear_abort(True)
        ctx.sync()
    finally:
        ctx.close()


def test_fused_sweep_wait_gives_up_is_reported_and_clears():
    """the one-launch red+black sweep of 513-point rows ("relax3d.fused" = 1): same protocol, same give-up path"""
    n3, dtype = (513, 129, 129), np.float64
    v, f = _data(n3, dtype, seed=5)
    ctx = P.Context(0)
    try:
        ctx.set_param("relax3d.fused", 1)
        ctx.set_param("relax3d.resident", 0)
        ctx.set_param("sync.spin_limit", 5000)
        ctx.set_param("test.handoff_fault", 3)
        t0 = time.time()
        P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        assert ctx.last_relax_kernel().startswith("sweep3d_xs_kernel"), ctx.last_relax_kernel()
        with pytest.raises(P.MgxError, match="gave up"):
            ctx.sync()
        assert time.time() - t0 < 5.0
        ctx.set_param("test.handoff_fault", 0)
        ctx.clear_abort(False)
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        assert not ctx.last_relax_kernel().startswith("sweep3d_xs_kernel"), ctx.last_relax_kernel()
        ctx.sync()
        want = O.relax3d(n3, RG, v, f, 2, dtype=dtype)
        assert bits_equal(got, want)
        ctx.clear_abort(True)
        got = P.ops3dxs.relax_pp(ctx, v, f, n3, RG, 2)
        assert ctx.last_relax_kernel().startswith("sweep3d_xs_kernel"), ctx.last_relax_kernel()
        ctx.sync()
        assert bits_equal(got, want)
    finally:
        ctx.close()


def test_shared_gpu_switch_keeps_the_kernels_off():
    n3, dtype = (65, 65, 65), np.float64
    v, f = _data(n3, dtype, seed=9)
    ctx = P.Context(0)
    try:
        ctx.set_param("gpu.exclusive", 0)
        ctx.set_param("relax3d.fused", 1)
        got = P.ops3dxs.relax(ctx, v, f, n3, RG, 5)
        assert not ctx.last_relax_kernel().startswith("relax3d_xs_resident"), ctx.last_relax_kernel()
        ctx.sync()
        assert bits_equal(got, O.relax3d(n3, RG, v, f, 5, dtype=dtype))
        with pytest.raises(P.MgxError):
            ctx.set_param("gpu.exclusive", 2)
        with pytest.raises(P.MgxError):
            ctx.set_param("sync.spin_limit", 0)
    finally:
        ctx.close()


def test_two_contexts_running_resident_relax_side_by_side():
    """two contexts of one process launch the resident kernel on 129^3 (256 workgroups of one per CU each) at the same time from
    two host threads -- the situation the exclusive-GPU assumption excludes.  Whatever the hardware makes of it, the contract
    holds: every launch terminates, and each context either has the right result or reports the given-up wait (and then
    computes the right result after clear_abort)."""
    n3, ncycles, dtype = (129, 129, 129), 40, np.float32
    v, f = _data(n3, dtype, seed=11)
    want = O.relax3d(n3, [0, 1, 0, 1, 0, 1], v, f, ncycles, dtype=dtype)
    out = {}

    def work(k):
        ctx = P.Context(0)
        try:
            ctx.set_param("sync.spin_limit", 200000)  # ~0.2 s: a deadlock between the two grids ends quickly
            res = []
            for _ in range(4):
                got = P.ops3dxs.relax(ctx, v, f, n3, [0, 1, 0, 1, 0, 1], ncycles)
                try:
                    ctx.sync()
                    res.append(("ok", bits_equal(got, want)))
                except P.MgxError:
                    ctx.clear_abort(False)
                    got = P.ops3dxs.relax(ctx, v, f, n3, [0, 1, 0, 1, 0, 1], ncycles)
                    ctx.sync()
                    res.append(("gave up, colour passes", bits_equal(got, want)))
            out[k] = res
        except Exception as e:  # noqa: BLE001
            out[k] = [("exception: %r" % (e,), False)]
        finally:
            ctx.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    t0 = time.time()
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert all(not t.is_alive() for t in ts), "a launch did not terminate"
    assert time.time() - t0 < 60
    for k in range(2):
        assert all(ok for _, ok in out[k]), out
