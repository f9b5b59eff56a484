"""GPU suite: the COMMUNICATION-AVOIDING slab schedule (csrc/host/mg_dist3d.inc, ca_*): slabs of at least ca_min_planes
planes carry 6 ghost planes either side and exchange v once per Relax call -- the first ghost planes are relaxed
redundantly, the region a rank may trust shrinking by one plane per colour pass -- instead of once per colour pass.
The cycle being distributed is N3/MultiGrid3D.cpp:623-647 (the reference is single-device).  Thread-ranks on the
asynchronous in-process transport with its delay hook on (tests/test_gpu_dist.py), every word of the result against the
oracle: a ghost plane that is trusted one pass too long, an edge / interior range that is off by a plane or an exchange
the compute stream does not wait for changes bits.  Exchanges are counted (mgDistMultiGrid3D::n_exchanges)."""
import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import bits_equal
from test_gpu_dist import run_ranks

pytestmark = pytest.mark.gpu
RG = [-1, 1, 0, 2, 0.5, 3]


def _data(n3, dtype, seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-1, 1, O.shape(n3)).astype(dtype), rng.uniform(-1, 1, O.shape(n3)).astype(dtype)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("inline_bytes", [0, None], ids=["overlapped", "inline"])
@pytest.mark.parametrize("nranks,n3,v1,v2,dtype", [
    (2, [65, 65, 129], 2, 2, np.float64),    # 64 / 32 / 16 planes per rank: three CA levels, the last one too thin to split
    (4, [65, 65, 257], 2, 2, np.float64),    # 64 / 32 / 16, middle ranks with two edges
    (4, [65, 33, 257], 1, 1, np.float32),    # one sweep: 2 planes deep
    (2, [65, 65, 129], 3, 3, np.float64),    # six passes = all six ghost planes in one go
    (2, [65, 33, 129], 4, 1, np.float64),    # eight passes: two chunks with an exchange in between
    (4, [33, 65, 257], 0, 2, np.float64),    # no pre-smoothing: the residual asks for its two planes itself
    (2, [65, 65, 129], 2, 0, np.float64),    # no post-smoothing: the plain correction + exchange of the other schedule
    (8, [33, 33, 257], 2, 2, np.float32),    # 32 / 16 planes on 8 ranks
])
def test_ca_vcycle_matches_oracle(nranks, n3, v1, v2, dtype, inline_bytes):
    v0, f0 = _data(n3, dtype, seed=sum(n3) + nranks + v1)
    counts = {}
    got, info = run_ranks(nranks, n3, RG, dtype, v1, v2, 2, 16, v0=v0, f0=f0, inline_bytes=inline_bytes, counts=counts)
    want = O.cycle3d(n3, RG, mode=0, v1=v1, v2=v2, reps=2, v=v0, f=f0, dtype=dtype)
    assert not np.isnan(got).any()
    assert bits_equal(got, want)
    assert all(counts[r] == counts[0] for r in range(nranks)), counts  # every rank takes part in every exchange


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nranks,n3", [(2, [257, 129, 257]), (4, [257, 129, 257])])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ca_wide_rows_fused_forms(nranks, n3, dtype):
    """rows of 257 points: the pipelined smoother, the correcting red pass on edge / interior / ghost planes (the plane below
    its range is NOT the grid's boundary plane on a slab) and, with "rr3d.black" = 2, the last black pass of the interior
    inside the residual+restrict launch while the exchange behind the pre-smoothing is in flight"""
    v0, f0 = _data(n3, dtype, seed=7 + nranks)
    names, counts = {}, {}
    got, info = run_ranks(nranks, n3, RG, dtype, 2, 2, 2, 16, v0=v0, f0=f0, inline_bytes=0, params={"rr3d.black": 2}, counts=counts,
                          extra=lambda mg: names.setdefault(mg.rank, (mg.ctx.last_rr_kernel(), mg.ctx.last_corr_kernel())))
    want = O.cycle3d(n3, RG, mode=0, v1=2, v2=2, reps=2, v=v0, f=f0, dtype=dtype)
    assert bits_equal(got, want)
    if dtype == np.float64:  # the fused way down is the fp64 kernel
        assert all(names[r][0].startswith("relax_rr3d_xs_kernel") for r in range(nranks)), names
    if (n3[2] - 1) // nranks >= 96:  # thinner slabs correct the black points in place with one launch and run plain passes
        assert all(names[r][1].startswith("relax3d_xs_pipe") for r in range(nranks)), names
    else:
        assert all(names[r][1] == "" for r in range(nranks)), names


@pytest.mark.timeout(300)
def test_ca_exchanges_per_cycle_are_counted_and_few():
    """V(2,2), 3 distributed levels + the replicated tail: per level one exchange of v behind the pre-smoothing and one behind
    the post-smoothing, one of f on the two coarser levels, the all-gather: at most 9 -- against one per colour pass and
    level before (about 30).  Same bits either way."""
    n3, nranks = [65, 65, 257], 4
    v0, f0 = _data(n3, np.float64, seed=11)
    res = {}
    for ca in (0, 16):
        counts = {}
        got, info = run_ranks(nranks, n3, RG, np.float64, 2, 2, 3, 16, v0=v0, f0=f0, inline_bytes=0, ca_min_planes=ca, counts=counts)
        assert info[0][0] == 3
        res[ca] = (got, counts[0])
    assert bits_equal(res[0][0], res[16][0])
    assert bits_equal(res[16][0], O.cycle3d(n3, RG, mode=0, v1=2, v2=2, reps=3, v=v0, f=f0, dtype=np.float64))
    assert max(res[16][1]) <= 9, res[16][1]
    assert min(res[0][1]) >= 25, res[0][1]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nranks", [2, 4])
def test_ca_fmg_relax_and_norm(nranks):
    """FullMultiGridVCycle (:569-585: Restrict of f, plain Interpolate), the public Relax and the residual norm on levels that
    run the communication-avoiding schedule: every reader of a ghost plane asks for it first"""
    n3 = [65, 65, 129]
    got, info = run_ranks(nranks, n3, RG, np.float64, 2, 2, 1, 16, fmg=1, inline_bytes=0)
    want = O.cycle3d(n3, RG, mode=1, v0=1, v1=2, v2=2, dtype=np.float64)
    want = O.cycle3d(n3, RG, mode=0, v1=2, v2=2, reps=1, v=want, dtype=np.float64)
    assert bits_equal(got, want)

    norms = {}

    def extra(mg):
        mg.Relax(0, 3)
        norms[mg.rank] = mg.ResidualNorm(0)
        mg.Relax(0, 1)
        return None

    v0, f0 = _data(n3, np.float64, seed=5)
    got, info = run_ranks(nranks, n3, RG, np.float64, 2, 2, 1, 16, v0=v0, f0=f0, inline_bytes=0, extra=extra)
    w = O.cycle3d(n3, RG, mode=0, v1=2, v2=2, reps=1, v=v0, f=f0, dtype=np.float64)
    w = O.relax3d(n3, RG, w, f0, 3, dtype=np.float64)
    ref = float(np.sqrt(np.sum(O.residual3d(n3, RG, w, f0, P.REF_COMPAT, np.float64) ** 2)))
    w = O.relax3d(n3, RG, w, f0, 1, dtype=np.float64)
    assert bits_equal(got, w)
    assert all(norms[r] == norms[0] for r in range(nranks))
    assert abs(norms[0] - ref) <= 1e-12 * ref
