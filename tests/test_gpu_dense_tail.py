"""Collapsed coarse levels (dense_op_gemv_kernel): the V-cycle on the levels >= l_c applied as ONE dense operator that
amgx_create forms with the device's own sub-cycle.  Checked against the oracle (amg_matrix.cpp:183-302 restated) and
against the same handle built with separate launches (AMGX_NO_DENSE_TAIL=1); both launch forms must stay green."""
import os

import numpy as np
import pytest

from tests.problems import poisson_case, elasticity_case, rhs

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _dev(H, dense=True, **kw):
    from ngsamg_amd.device import DeviceAMGMatrix
    old = os.environ.pop("AMGX_NO_DENSE_TAIL", None)
    try:
        if not dense:
            os.environ["AMGX_NO_DENSE_TAIL"] = "1"
        return DeviceAMGMatrix(H, device=0, **kw)
    finally:
        os.environ.pop("AMGX_NO_DENSE_TAIL", None)
        if old is not None:
            os.environ["AMGX_NO_DENSE_TAIL"] = old


@pytest.mark.parametrize("shape,diri,mcs", [((33, 33), "left|top", 5), ((17, 17, 17), "right|top", 20), ((9, 30, 13), ".*", 20), ((31, 31, 31), "right|top", 10)])
@pytest.mark.parametrize("sm,osm,tol", [("jacobi", "jacobi", 1e-12), ("gs", "gs_mc", 1e-10)])
def test_dense_tail_matches_oracle_and_separate_launches(shape, diri, mcs, sm, osm, tol):
    from oracle.pyoracle import Oracle
    p, H = poisson_case(shape, diri, mcs)
    b = rhs(p, 11)
    ref = Oracle(H.levels, sm_type=osm).apply(b)
    d1, d0 = _dev(H, True, sm_type=sm), _dev(H, False, sm_type=sm)
    ci1, ci0 = d1.cycle_info(), d0.cycle_info()
    assert ci0["dense_level"] == -1
    if H.n_levels >= 3:
        assert 1 <= ci1["dense_level"] <= H.n_levels - 2 and ci1["dense_n"] == H.levels[ci1["dense_level"]].n
        assert ci1["tail_level"] == -1
    x1, x0 = np.full(p.n, np.nan), np.full(p.n, np.nan)
    d1.Mult(b, x1)
    d0.Mult(b, x0)
    assert _rel(x1, ref) < tol and _rel(x0, ref) < tol
    assert _rel(x1, x0) < 1e-13
    # the work vectors of the collapsed levels are clean after the build: a second application gives the same result
    x2 = np.full(p.n, np.nan)
    d1.Mult(b, x2)
    assert np.array_equal(x1, x2)


def test_dense_tail_hybrid_gs_and_symmetric_steps():
    """block-hybrid Gauss-Seidel levels and the ProxySmoother wrapper (sm_steps = 2, sm_symm) below the collapse point"""
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    p, H = poisson_case((31, 31, 31), "right|top", 10)
    b = rhs(p, 5)
    d1, d0 = _dev(H, True, sm_type="hgs"), _dev(H, False, sm_type="hgs")
    assert d1.cycle_info()["dense_level"] >= 1
    lv, types = hgs_levels(H.levels, d1.hgs)
    ref = Oracle(lv, sm_type=types).apply(b)
    x1, x0 = np.empty(p.n), np.empty(p.n)
    d1.Mult(b, x1)
    d0.Mult(b, x0)
    assert _rel(x1, ref) < 1e-10 and _rel(x1, x0) < 1e-13
    for kw in ({"sm_steps": 2}, {"sm_symm": True}):
        d1, d0 = _dev(H, True, sm_type="jacobi", **kw), _dev(H, False, sm_type="jacobi", **kw)
        ref = Oracle(H.levels, sm_type="jacobi", **kw).apply(b)
        d1.Mult(b, x1)
        d0.Mult(b, x0)
        assert d1.cycle_info()["dense_level"] >= 1
        assert _rel(x1, ref) < 1e-12 and _rel(x1, x0) < 1e-13


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("sm,osm", [("jacobi", "jacobi"), ("gs", "gs_mc")])
def test_dense_tail_block_levels(rot, sm, osm):
    from oracle.pyoracle import Oracle
    p, H = elasticity_case((9, 9, 9), rotations=rot, max_coarse_size=5)
    b = rhs(p, 2)
    ref = Oracle(H.levels, sm_type=osm).apply(b)
    d1, d0 = _dev(H, True, sm_type=sm), _dev(H, False, sm_type=sm)
    ci = d1.cycle_info()
    if H.n_levels >= 3:
        assert ci["dense_level"] >= 1 and ci["dense_n"] == H.levels[ci["dense_level"]].n * H.levels[ci["dense_level"]].bs
    x1, x0 = np.empty(p.n * p.bs), np.empty(p.n * p.bs)
    d1.Mult(b, x1)
    d0.Mult(b, x0)
    assert _rel(x1, ref) < 1e-10 and _rel(x0, ref) < 1e-10 and _rel(x1, x0) < 1e-12


def test_dense_tail_cap_and_other_cycles():
    """AMGX_DENSE_MAX caps the operator; W and BS cycles never use it (their sub-cycles are different operators)"""
    from ngsamg_amd.device import DeviceAMGMatrix
    p, H = poisson_case((31, 31, 31), "right|top", 10)
    os.environ["AMGX_DENSE_MAX"] = "50"
    try:
        d = DeviceAMGMatrix(H, device=0, sm_type="jacobi")
    finally:
        del os.environ["AMGX_DENSE_MAX"]
    ci = d.cycle_info()
    assert ci["dense_level"] == -1 or ci["dense_n"] <= 50
    for cyc in ("W", "BS"):
        assert DeviceAMGMatrix(H, device=0, sm_type="jacobi", mg_cycle=cyc).cycle_info()["dense_level"] == -1
