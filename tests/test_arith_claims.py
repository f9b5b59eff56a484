"""CPU suite: the two arithmetic identities the HIP kernels rely on to avoid IEEE divisions, checked with numpy on the host
(the kernels themselves are compared bit for bit with the oracle / the reference fixtures in the GPU suite).

  1. residual3d_point, MODE | 2 (mgx_kernels3d.hpp): for a power of two h2, t / h2 == t * (1 / h2), any t.
  2. relax3d_point_rd, fp32 (mgx_kernels3d.hpp): for fp32 num, den the quotient num / den equals
     float32(float64(num) * RN53(1 / float64(den))) whenever the result is a normal fp32 number (the kernel divides for
     real otherwise).  Random operands over the whole exponent range plus numerators placed next to the rounding
     boundaries of the quotient."""
import numpy as np

F32_MIN = np.float32(1.17549435e-38)


def test_division_by_power_of_two_is_multiplication_by_reciprocal():
    rng = np.random.default_rng(1)
    for dtype, emax in ((np.float32, 120), (np.float64, 1000)):
        t = (rng.uniform(-2, 2, 1 << 20) * 2.0 ** rng.integers(-emax - 40, emax, 1 << 20)).astype(dtype)
        t[:4] = [0.0, -0.0, np.inf, np.finfo(dtype).tiny / 4]
        for k in (0, 2, 8, 18, 20, 22, 60):
            h2 = dtype(2.0) ** dtype(-k)
            with np.errstate(over="ignore", under="ignore"):
                assert np.array_equal((t / h2).view(np.uint8), (t * (dtype(1) / h2)).view(np.uint8)), (dtype, k)


def _check(num, den):
    with np.errstate(over="ignore", under="ignore", divide="ignore", invalid="ignore"):
        want = num / den
        got = (num.astype(np.float64) * (1.0 / den.astype(np.float64))).astype(np.float32)
    ok = np.abs(got) >= F32_MIN  # the kernel's guard: everything else is divided for real
    assert np.array_equal(got[ok].view(np.uint32), want[ok].view(np.uint32))
    return int(ok.sum())


def test_fp32_quotient_through_a_double_reciprocal():
    rng = np.random.default_rng(2)
    n = 1 << 22
    num = (rng.uniform(-2, 2, n) * 2.0 ** rng.integers(-100, 100, n)).astype(np.float32)
    den = (rng.uniform(1, 2, n) * 2.0 ** rng.integers(-60, 60, n)).astype(np.float32)
    assert _check(num, den) > n // 2
    # denominators of the smoother: 2 (hy2 hz2 + hx2 hz2 + hx2 hy2) for 2^k + 1 points on boxes of odd shapes
    for k in range(1, 11):
        for box in ((1, 1, 1), (2, 1, 0.5), (3, 1.7, 0.3)):
            hx2, hy2, hz2 = [np.float32(np.float32(b) / np.float32(2 ** k)) ** 2 for b in box]
            d = np.float32(2) * (hy2 * hz2 + hx2 * hz2 + hx2 * hy2)
            _check(num, np.full(n, d, np.float32))
    # numerators next to the rounding boundaries of the quotient: num = RN(den * (q + ulp(q) / 2)) and its neighbours
    q = rng.uniform(1, 2, n).astype(np.float32)
    mid = q.astype(np.float64) + 2.0 ** -24
    base = (den.astype(np.float64) * mid).astype(np.float32)
    for step in (-1, 0, 1):
        _check((base.view(np.int32) + step).view(np.float32), den)
