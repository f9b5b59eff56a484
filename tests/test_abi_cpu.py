"""CPU suite: the C-ABI library loads, exports every symbol include/*.h declares, reports errors
as status codes, and fails loudly (no fallback) when no GPU is present.  The 1D host-only path
(BASELINE.json configs[0]) is checked against the oracle and the golden fixtures here because
it needs no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
import pde_multigrid_amd as P
from conftest import ROOT, bits_equal, load_golden


def _declared_symbols():
    syms = set()
    for hdr, stamp in (("mgx.h", "MGX_DECLARE_OPS"), ("mg_multigrid.h", "MG_DECLARE")):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        # split the stamping macro body from the plain declarations
        m = re.search(r"#define %s\(\w+, real\)(.*?)\n\n" % stamp, text, flags=re.S)
        body = m.group(1).replace("\\\n", "\n")
        plain = text.replace(m.group(0), "")
        for name in re.findall(r"\b(?:int|void|const char\*)\s+(mgx?\w+)\s*\(", plain):
            syms.add(name)
        for name in re.findall(r"\b(?:int|void)\s+(mg\w*##\w+(?:##\w+)*)\s*\(", body):
            for sfx in ("f32", "f64"):
                syms.add(name.replace("##SFX", sfx).replace("##R##", sfx).replace("##R", sfx))
    return sorted(syms)


def test_library_exports_every_declared_symbol():
    syms = _declared_symbols()
    assert len(syms) > 120, len(syms)
    missing = [s for s in syms if not hasattr(P.lib, s)]
    assert not missing, missing


def test_no_gpu_fails_loudly_or_gpu_present():
    n = C.c_int(-1)
    st = P.lib.mgx_device_count(C.byref(n))
    if st == P.MGX_OK and n.value > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(P.MgxError) as e:
        P.Context(0)
    assert e.value.status == P.MGX_ERR_NOGPU
    assert "no CPU fallback" in str(e.value)


def test_status_strings_and_null_arguments():
    assert P.status_string(0) == "MGX_OK"
    assert P.status_string(2) == "MGX_ERR_SIZE"
    assert P.lib.mgx_ctx_sync(None) == P.MGX_ERR_INVALID
    assert P.lib.mgx3d_relax_f64(None, None, None, None, None, 1) == P.MGX_ERR_INVALID
    assert b"NULL" in P.lib.mgx_last_error()


def test_level_rule_matches_reference():
    for k in range(2, 12):
        assert P.num_grids(2 ** k + 1) == k == O.num_grids(2 ** k + 1)
    assert P.coarse_size((513, 257, 5)) == (257, 129, 3)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "pde_multigrid_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hpp", ".hip", ".inc")):
                src = open(os.path.join(dp, fn)).read()
                assert "mg_oracle" not in src and "import oracle" not in src and "libmgoracle" not in src, fn


# ------------------------------------------------------------------ 1D host path (configs[0])
@pytest.mark.parametrize("n", [17, 65])
def test_1d_ops_vs_golden(n):
    g = load_golden("ops1d_n%d.npz" % n)
    mg = P.MultiGrid1D(n, g["range"].tolist(), np.float32)
    v, f, c = g["v"], g["f"], g["c"]
    for k, key in ((1, "relax1"), (3, "relax3")):
        mg.v()[:] = v
        mg.f()[:] = f
        mg.Relax(0, k)
        assert bits_equal(mg.v().copy(), g[key])
    mg.v()[:] = v
    mg.f()[:] = f
    assert bits_equal(mg.CalculateResidual(0), g["residual"])
    assert bits_equal(mg.Restrict(v), g["restrict"])
    assert bits_equal(mg.Interpolate(v, c), g["interpolate"])
    assert bits_equal(mg.ApplyCorrection(v, f), g["correct"])
    assert bits_equal(mg.setToValue(v, 2.5, False), g["set_interior"])
    assert bits_equal(mg.setToValue(v, 2.5, True), g["set_all"])
    assert bits_equal(P.solve1d(v, f, g["range"].tolist(), ncycles=1), g["vcycle22"])
    assert bits_equal(P.solve1d(v, f, g["range"].tolist(), fmg=True, v0=1), g["fmg122"])


def test_1d_baseline_config0(known_answers):
    """BASELINE.json configs[0]: 1D, 4097 points, 5-level V(2,2) on the CPU path."""
    ka = known_answers["1d_n4097_vcycle22_5lev"]
    mg = P.MultiGrid1D(4097, [0, 1], np.float32, nlevels=5)
    assert mg.maxGrids == 12 and mg.numGrids == 5
    mg.VCycle(0, 2, 2)
    v = mg.v().copy()
    assert O.fnv(v) == ka["hash"]
    assert float(v[2048]) == ka["centre"]


@pytest.mark.parametrize("name", ["1d_n4097_fmg122", "1d_n257_fmg_2_1000_1000"])
def test_1d_fmg_known_answers(known_answers, name):
    ka = known_answers[name]
    mg = P.MultiGrid1D(ka["n"], [0, 1], np.float32)
    mg.FullMultiGridVCycle(0, ka["v0"], ka["v1"], ka["v2"])
    assert O.fnv(mg.v().copy()) == ka["hash"]


def test_1d_f64_matches_oracle_bitwise():
    mg = P.MultiGrid1D(1025, [0, 1], np.float64, nlevels=4)
    mg.VCycle(0, 2, 2)
    assert bits_equal(mg.v().copy(), O.cycle1d(1025, [0, 1], nlevels=4, mode=0, dtype=np.float64))


def test_1d_errors_are_status_codes():
    with pytest.raises(P.MgxError) as e:
        P.MultiGrid1D(100, [0, 1])
    assert e.value.status == P.MGX_ERR_SIZE
    mg = P.MultiGrid1D(17, [0, 1])
    with pytest.raises(P.MgxError) as e:
        mg._call("Restrict", np.zeros(17, np.float32).ctypes.data_as(C.c_void_p), C.c_int(17),
                 np.zeros(8, np.float32).ctypes.data_as(C.c_void_p), C.c_int(8))
    assert e.value.status == P.MGX_ERR_SIZE
    with pytest.raises(ValueError):
        mg.numGrids = 9
