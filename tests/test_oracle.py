"""CPU oracle: pinned by the committed golden fixtures, by the structural invariants the reference itself
asserts (symmetric smoothing => symmetric preconditioner, amg_pc.cpp:162-173) and by the iteration budgets
of the reference's pytest suite as sanity bounds."""
import numpy as np
import pytest

from oracle.pyoracle import Oracle
from tests import golden_io
from tests.problems import poisson_case, elasticity_case, rhs


@pytest.mark.parametrize("name", golden_io.NAMES)
@pytest.mark.parametrize("sm", ["jacobi", "gs", "gs_mc"])
def test_oracle_reproduces_golden(name, sm):
    z, levels = golden_io.load(name)
    b = z["b"]
    orc = Oracle(levels, sm_type=sm)
    x = np.zeros_like(b)
    r = b.copy()
    orc.smooth(0, x, b, r, True, True, True)
    assert np.allclose(x, z[f"{sm}_x_pre"], rtol=0, atol=1e-13 * np.abs(z[f"{sm}_x_pre"]).max())
    assert np.allclose(r, z[f"{sm}_r_pre"], rtol=0, atol=1e-13 * np.abs(b).max())
    if len(levels) > 1:
        assert np.allclose(orc.transfer_f2c(0, r), z[f"{sm}_b1"], rtol=0, atol=1e-13 * np.abs(b).max())
    for cyc in ("V", "W", "BS"):
        got = Oracle(levels, sm_type=sm, cycle=cyc).apply(b)
        ref = z[f"{sm}_{cyc}"]
        assert np.linalg.norm(got - ref) <= 1e-13 * np.linalg.norm(ref)
    got = Oracle(levels, sm_type=sm, sm_steps=2, sm_symm=True).apply(b)
    assert np.linalg.norm(got - z[f"{sm}_V_symm2"]) <= 1e-13 * np.linalg.norm(z[f"{sm}_V_symm2"])


@pytest.mark.parametrize("sm", ["jacobi", "gs", "gs_mc"])
@pytest.mark.parametrize("cycle", ["V", "W"])
def test_preconditioner_is_symmetric(sm, cycle):
    p, H = poisson_case((17, 17, 17), "right|top", 20)
    orc = Oracle(H.levels, sm_type=sm, cycle=cycle)
    u, v = rhs(p, 1), rhs(p, 2)
    a, b = np.dot(orc.apply(u), v), np.dot(u, orc.apply(v))
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b))


def test_res_form_equals_rhs_form():
    """GS keeping the residual current (row-transpose scatter) == GS in gather form + residual SpMV for
    symmetric A (the identity the GPU kernels rely on)."""
    p, H = poisson_case((33, 33), "left|top", 5)
    A = H.levels[0].A.to_scipy()
    orc = Oracle(H.levels, sm_type="gs")
    b = rhs(p, 3)
    rng = np.random.default_rng(4)
    for back in (False, True):
        x0 = rng.standard_normal(p.n) * p.free
        x1, r1 = x0.copy(), b - A @ x0
        orc.smooth(0, x1, b, r1, True, True, False, back)          # RES form
        x2, r2 = x0.copy(), np.zeros(p.n)
        orc.smooth(0, x2, b, r2, False, False, False, back)        # RHS form
        assert np.linalg.norm(x1 - x2) < 1e-12 * np.linalg.norm(x2)
        assert np.linalg.norm(r1 - (b - A @ x2)) < 1e-11 * np.linalg.norm(b)


def test_multadd_and_single_stage_consistency():
    p, H = poisson_case((17, 17, 17), "right|top", 20)
    orc = Oracle(H.levels, sm_type="jacobi")
    b = rhs(p, 5)
    y = np.ones(p.n)
    orc.apply_add(-2.0, b, y)
    assert np.allclose(y, 1.0 - 2.0 * orc.apply(b), rtol=1e-13, atol=1e-13)
    # Jacobi V(1,1) by hand on a 2-level hierarchy slice
    L0 = H.levels[0]
    A = L0.A.to_scipy()
    x = 0.9 * L0.dinv * b
    r = b - A @ x
    xs, rs = np.zeros(p.n), b.copy()
    orc.smooth(0, xs, b, rs, True, True, True)
    assert np.allclose(xs, x, rtol=1e-14, atol=1e-14) and np.allclose(rs, r, rtol=1e-12, atol=1e-12)


def test_iteration_budgets_as_sanity_bounds():
    """Reference pins: 2D P1 Poisson CG < 30 its at tol 1e-12 on a ~600-DOF netgen mesh (tests/h1/simple/
    test_2d_lo.py); the build's own hierarchy on a 50k-DOF grid must stay in the same league."""
    p, H = poisson_case((224, 224), "left|top", 5)
    for sm, budget in (("gs", 60), ("gs_mc", 60), ("jacobi", 70)):
        _, it, errs = Oracle(H.levels, sm_type=sm).pcg(p.load, tol=1e-12, maxit=100)
        assert it < budget and errs[-1] < 1e-12 * errs[0]
    p, H = elasticity_case((9, 9, 9), False, 10)
    _, it, errs = Oracle(H.levels, sm_type="gs").pcg(p.load, tol=1e-6, maxit=100)
    assert it < 60 and errs[-1] < 1e-6 * errs[0]
