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


@pytest.mark.parametrize("name", golden_io.NAMES)
@pytest.mark.parametrize("sm", ["jacobi", "gs", "gs_mc", "bgs", "bgs_mc"])
def test_oracle_pcg_history_reproduces_golden(name, sm):
    """PCG residual histories (err_k = sqrt(<C r_k, r_k>), tol 1e-8) pinned per smoother kind"""
    z, levels = golden_io.load(name)
    orc = Oracle(levels, sm_type=sm, bgs=[L.bgs for L in levels])
    _, it, errs = orc.pcg(z["load"], tol=1e-8, maxit=100)
    ref = z[f"{sm}_pcg_errs"]
    assert it + 1 == ref.size
    assert np.allclose(errs, ref, rtol=1e-7, atol=1e-9 * ref[0])     # late iterates amplify rounding differences


@pytest.mark.parametrize("name", golden_io.NAMES)
@pytest.mark.parametrize("sm", ["bgs", "bgs_mc"])
def test_oracle_block_gs_reproduces_golden(name, sm):
    """block Gauss-Seidel pins: block tables and inverses come from the fixture, not from a fresh setup"""
    z, levels = golden_io.load(name)
    b = z["b"]
    bgs = [L.bgs for L in levels]
    orc = Oracle(levels, sm_type=sm, bgs=bgs)
    x, r = np.zeros_like(b), b.copy()
    orc.smooth(0, x, b, r, True, True, True)
    assert np.allclose(x, z[f"{sm}_x_pre"], rtol=0, atol=1e-13 * np.abs(z[f"{sm}_x_pre"]).max())
    assert np.allclose(r, z[f"{sm}_r_pre"], rtol=0, atol=1e-13 * np.abs(b).max())
    for cyc in ("V", "W"):
        got = Oracle(levels, sm_type=sm, cycle=cyc, bgs=bgs).apply(b)
        assert np.linalg.norm(got - z[f"{sm}_{cyc}"]) <= 1e-13 * np.linalg.norm(z[f"{sm}_{cyc}"])


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


# ---- block Gauss-Seidel (reference BSmoother, block_gssmoother.cpp) ---------------------------------------------------

def _bgs_case(kind):
    if kind == "poisson":
        p, H = poisson_case((12, 12, 12), "right|top", 20)
    elif kind == "elast3":
        p, H = elasticity_case((9, 8, 7), False, 5, 0.12)
    else:
        p, H = elasticity_case((8, 7, 6), True, 5, 0.12)
    return p, H, H.build_bgs()


@pytest.mark.parametrize("kind", ["poisson", "elast3", "elast6"])
def test_bgs_blocks_inverses_and_colouring(kind):
    """host data of the block smoother: blocks = aggregates (disjoint, cover the free rows), Dinv_B A_BB = I (or the
    Moore-Penrose identities with pinv), coupled blocks never share a colour"""
    p, H, bgs = _bgs_case(kind)
    for lv, g in zip(H.levels[:-1], bgs[:-1]):
        A = lv.A.to_scipy().toarray() if lv.n * lv.bs <= 3000 else None
        rows = np.sort(g.block_rows)
        assert np.array_equal(rows, np.nonzero(lv.free & (lv.agg >= 0))[0])
        blockof = -np.ones(lv.n, dtype=int)
        for k in range(g.n_blocks):
            r = g.block_rows[g.block_ptr[k]:g.block_ptr[k + 1]]
            assert np.all(np.diff(r) > 0) and np.all(lv.agg[r] == lv.agg[r[0]])
            blockof[r] = k
            if A is not None:
                idx = (r[:, None] * lv.bs + np.arange(lv.bs)[None, :]).ravel()
                M = idx.size
                D = g.dinv[g.dinv_ptr[k]:g.dinv_ptr[k + 1]].reshape(M, M).T          # column-major
                AB = A[np.ix_(idx, idx)]
                assert np.abs(AB @ D @ AB - AB).max() <= 1e-9 * np.abs(AB).max()
        rr = np.repeat(np.arange(lv.A.n_rows), np.diff(lv.A.rowptr))        # block-row index of every stored block
        bi, bj = blockof[rr], blockof[lv.A.col]
        m = (bi >= 0) & (bj >= 0) & (bi != bj)
        assert not np.any(g.color[bi[m]] == g.color[bj[m]])
        assert g.color.max() + 1 == g.n_colors


@pytest.mark.parametrize("kind", ["poisson", "elast3"])
def test_bgs_res_form_equals_rhs_form_and_flags(kind):
    """BSmoother::Smooth (block_gssmoother.cpp:434-498): update_res -> RES form (row-transpose scatter), else RHS form;
    for symmetric A both give the same x and the RES form keeps res = b - A x"""
    p, H, bgs = _bgs_case(kind)
    n = p.n * H.levels[0].bs
    A = H.levels[0].A.to_scipy()
    rng = np.random.default_rng(3)
    b = rhs(p, 2)
    for sm in ("bgs", "bgs_mc"):
        orc = Oracle(H.levels, sm_type=sm, bgs=bgs)
        for back in (False, True):
            x0 = rng.standard_normal(n)
            xa, ra = orc.smooth(0, x0.copy(), b, np.zeros(n), False, True, False, back)
            xb, _ = orc.smooth(0, x0.copy(), b, np.zeros(n), False, False, False, back)
            r0 = b - A @ x0
            xc, rc = orc.smooth(0, x0.copy(), b, r0.copy(), True, True, False, back)
            assert np.linalg.norm(xa - xb) <= 1e-13 * np.linalg.norm(xb)
            assert np.linalg.norm(xc - xb) <= 1e-13 * np.linalg.norm(xb)
            assert np.linalg.norm(ra - (b - A @ xa)) <= 1e-12 * np.linalg.norm(b)
            assert np.linalg.norm(rc - ra) <= 1e-12 * np.linalg.norm(b)
        z, rz = orc.smooth(0, np.zeros(n), b, np.zeros(n), False, True, True)       # x_zero: res starts as b
        assert np.linalg.norm(rz - (b - A @ z)) <= 1e-12 * np.linalg.norm(b)


@pytest.mark.parametrize("kind", ["poisson", "elast3", "elast6"])
def test_bgs_cycle_is_symmetric_and_beats_point_gs(kind):
    """forward pre- / backward post-smoothing over the same block order => symmetric preconditioner (amg_pc.cpp:162-173);
    PCG needs no more iterations than with point Gauss-Seidel (why the reference recommends it for elasticity)"""
    p, H, bgs = _bgs_case(kind)
    n = p.n * H.levels[0].bs
    rng = np.random.default_rng(0)
    free = np.repeat(p.free, H.levels[0].bs)
    u, v = rng.standard_normal(n) * free, rng.standard_normal(n) * free
    for sm in ("bgs", "bgs_mc"):
        orc = Oracle(H.levels, sm_type=sm, bgs=bgs)
        a, b_ = float(orc.apply(u) @ v), float(u @ orc.apply(v))
        assert abs(a - b_) <= 1e-11 * max(abs(a), abs(b_))
    b = rhs(p, 1)
    _, it_b, e_b = Oracle(H.levels, sm_type="bgs", bgs=bgs).pcg(b, tol=1e-8, maxit=100)
    _, it_g, _ = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=100)
    assert e_b[-1] <= 1e-8 * e_b[0] and it_b <= it_g + 1


def test_oracle_gmres_minimises_the_preconditioned_residual():
    """orc_gmres (test infrastructure for the GMRES parity test): the recurrence value equals the true |C (b - A x)|, it is
    monotone inside a restart cycle, full GMRES needs no more iterations than PCG, and both reach the same solution"""
    from oracle.pyoracle import Oracle
    from tests.problems import poisson_case, rhs
    p, H = poisson_case((13, 13, 13), "right|top", 20)
    orc = Oracle(H.levels, sm_type="jacobi")
    b = rhs(p, 3)
    x, it, errs = orc.gmres(b, tol=1e-10, maxit=100, restart=40)
    xc, itc, _ = orc.pcg(b, tol=1e-10, maxit=100)
    assert errs[-1] <= 1e-10 * errs[0] and it <= itc + 1
    assert all(e2 <= e1 * (1 + 1e-12) for e1, e2 in zip(errs[:-1], errs[1:]))
    A = H.levels[0].A.to_scipy()
    assert abs(np.linalg.norm(orc.apply(b - A @ x)) - errs[-1]) <= 1e-6 * errs[0]
    assert np.linalg.norm(x - xc) <= 1e-7 * np.linalg.norm(xc)
    # restarted: same fixed point, more iterations
    x5, it5, errs5 = orc.gmres(b, tol=1e-10, maxit=300, restart=5)
    assert it5 >= it and np.linalg.norm(x5 - xc) <= 1e-7 * np.linalg.norm(xc)
    # without a preconditioner the first step minimises |b - alpha A b|
    _, _, e = orc.gmres(b, tol=0.0, maxit=1, restart=5, precond=False)
    Ab = A @ b
    assert abs(e[1] - np.linalg.norm(b - (Ab @ b) / (Ab @ Ab) * Ab)) <= 1e-10 * e[0]
