"""Container-only: oracle<float> against the UNMODIFIED compiled reference
(oracle/_ref/libmgref.so, `make -C oracle ref`).  Skipped where /root/reference (and
therefore the _ref build) does not exist, e.g. on the GPU box, where tests/golden/ pins
the oracle instead.  Bar: bit-exact on seeded random inputs with anisotropic ranges."""
import numpy as np
import pytest

import oracle as O
import refshim as R
from conftest import bits_equal

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (no /root/reference)")
A = [-1, -2, 0, -3]


@pytest.mark.parametrize("n,seed", [(5, 1), (9, 2), (17, 3), (33, 4)])
def test_3d_ops(n, seed):
    rng = np.random.default_rng(seed)
    n3, rg = [n] * 3, [-1, 1, 0, 2, 0.5, 3]
    v = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)
    f = rng.uniform(-1, 1, O.shape(n3)).astype(np.float32)
    c = rng.uniform(-1, 1, O.shape(O.csize(n3))).astype(np.float32)
    for k in (1, 2, 5):
        assert bits_equal(R.relax3d(n3, rg, v, f, k), O.relax3d(n3, rg, v, f, k))
    assert bits_equal(R.residual3d(n3, rg, v, f), O.residual3d(n3, rg, v, f))
    assert bits_equal(R.restrict3d(n3, v), O.restrict3d(n3, v))
    assert bits_equal(R.interpolate3d(n3, v, c), O.interpolate3d(n3, v, c))
    assert bits_equal(R.correct3d(n3, v, f), O.correct3d(n3, v, f))
    for b in (0, 1):
        assert bits_equal(R.set3d(n3, v, -7.25, b), O.set3d(n3, v, -7.25, b))
    assert bits_equal(R.cycle3d(n3, rg, mode=0, v1=3, v2=1, reps=2, v=v, f=f),
                      O.cycle3d(n3, rg, mode=0, v1=3, v2=1, reps=2, v=v, f=f))
    assert bits_equal(R.cycle3d(n3, rg, mode=1, v0=2, v1=1, v2=2, v=v, f=f),
                      O.cycle3d(n3, rg, mode=1, v0=2, v1=1, v2=2, v=v, f=f))


@pytest.mark.parametrize("n,nlev", [(17, 0), (33, 3), (65, 0)])
def test_3d_analytic_cycles(n, nlev):
    r = [0, 1, 0, 1, 0, 1]
    assert bits_equal(R.cycle3d([n] * 3, r, nlevels=nlev, mode=0), O.cycle3d([n] * 3, r, nlevels=nlev, mode=0))
    assert bits_equal(R.cycle3d([n] * 3, r, nlevels=nlev, mode=1), O.cycle3d([n] * 3, r, nlevels=nlev, mode=1))


@pytest.mark.parametrize("n,seed", [(9, 1), (17, 2), (65, 3)])
def test_2d_ops(n, seed):
    rng = np.random.default_rng(seed)
    n2, rg = [n] * 2, [0, 20, -3, 20]
    v = rng.uniform(-1, 1, O.shape(n2)).astype(np.float32)
    f = rng.uniform(-1, 1, O.shape(n2)).astype(np.float32)
    c = rng.uniform(-1, 1, O.shape(O.csize(n2))).astype(np.float32)
    for k in (1, 4):
        assert bits_equal(R.relax2d(n2, rg, A, 2, v, f, k), O.relax2d(n2, rg, A, 2, v, f, k))
    assert bits_equal(R.residual2d(n2, rg, A, 2, v, f), O.residual2d(n2, rg, A, 2, v, f))
    assert bits_equal(R.restrict2d(n2, v), O.restrict2d(n2, v))
    assert bits_equal(R.interpolate2d(n2, v, c), O.interpolate2d(n2, v, c))
    assert bits_equal(R.correct2d(n2, v, f), O.correct2d(n2, v, f))
    for b in (0, 1):
        assert bits_equal(R.set2d(n2, v, 3.5, b), O.set2d(n2, v, 3.5, b))
    assert bits_equal(R.cycle2d(n2, rg, A, 2, mode=0, reps=2, v=v, f=f), O.cycle2d(n2, rg, A, 2, mode=0, reps=2, v=v, f=f))
    assert bits_equal(R.cycle2d(n2, rg, A, 2, mode=1, v0=2, v=v, f=f), O.cycle2d(n2, rg, A, 2, mode=1, v0=2, v=v, f=f))


def test_2d_thesis_range_analytic():
    # thesis runs use [0,20]^2
    n2, rg = [129] * 2, [0, 20, 0, 20]
    assert bits_equal(R.cycle2d(n2, rg, A, 2, mode=1, v0=1, v1=20, v2=20), O.cycle2d(n2, rg, A, 2, mode=1, v0=1, v1=20, v2=20))


@pytest.mark.parametrize("n,seed", [(9, 1), (33, 2), (257, 3)])
def test_1d_ops(n, seed):
    rng = np.random.default_rng(seed)
    rg = [0, 1]
    v = rng.uniform(-1, 1, n).astype(np.float32)
    f = rng.uniform(-1, 1, n).astype(np.float32)
    c = rng.uniform(-1, 1, (n - 1) // 2 + 1).astype(np.float32)
    for k in (1, 4):
        assert bits_equal(R.relax1d(n, rg, v, f, k), O.relax1d(n, rg, v, f, k))
    assert bits_equal(R.residual1d(n, rg, v, f), O.residual1d(n, rg, v, f))
    assert bits_equal(R.restrict1d(n, v), O.restrict1d(n, v))
    assert bits_equal(R.interpolate1d(n, v, c), O.interpolate1d(n, v, c))
    assert bits_equal(R.correct1d(n, v, f), O.correct1d(n, v, f))
    assert bits_equal(R.cycle1d(n, rg, mode=0, reps=2, v=v, f=f), O.cycle1d(n, rg, mode=0, reps=2, v=v, f=f))
    assert bits_equal(R.cycle1d(n, rg, mode=1, v0=2, v=v, f=f), O.cycle1d(n, rg, mode=1, v0=2, v=v, f=f))
