"""The reference's own pytest drivers, re-hosted: ``Preconditioner(a, "ngs_amg.h1_scal", **flags)`` + CG with
``assert cg.errors[-1] < tol*cg.errors[0]`` and ``assert cg.iterations < ms`` (reference tests/h1/amg_utils.py:337-363,
tests/h1/simple/test_2d_lo.py, tests/h1/simple/test_vec.py, tests/elasticity/mdim/simple/test_3d_lo.py).
Meshes are the build's structured grids instead of netgen meshes, so the budgets are the build's own."""
import numpy as np
import pytest

from ngsamg_amd import fem, Matrix, NgsAMGError

pytestmark = pytest.mark.gpu


def Solve(mat, rhs, c, ms=100, tol=1e-12):
    """the reference harness (tests/h1/amg_utils.py:337-363) re-hosted in ngsamg_amd.harness"""
    from ngsamg_amd.harness import Solve as _Solve
    return _Solve(c, rhs, ms=ms, tol=tol, quiet=True)


def _mat(p):
    return Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)


def test_2d_lo_reference_size_and_budget():
    """reference tests/h1/simple/test_2d_lo.py: unit square, maxh = 0.05 (~500 vertices), max_coarse_size 5, tol 1e-12,
    ms = 30 -- same size class (23 x 23 vertices), same budget"""
    from ngsamg_amd import ngs_amg
    p = fem.poisson_fast((23, 23), dirichlet="left|top")
    c = ngs_amg.Preconditioner(_mat(p), "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2)
    Solve(_mat(p), p.load, c, ms=30)


def test_2d_lo():
    """the same problem 20x larger than the reference's test mesh, still inside the reference's budget of 30"""
    from ngsamg_amd import ngs_amg
    p = fem.poisson_fast((101, 101), dirichlet="left|top")
    c = ngs_amg.Preconditioner(_mat(p), "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2)
    sol, cg = Solve(_mat(p), p.load, c, ms=30)               # the reference's budget (SPW hierarchy: 26-27 iterations)
    A = p.to_scipy()
    f = p.free.astype(bool)
    assert np.linalg.norm((A @ sol - p.load)[f]) < 1e-9 * np.linalg.norm(p.load)
    assert c.GetNLevels() >= 3 and c.GetBlockSize(0) == 1 and c.GetNDof(0) == p.n
    assert 1.0 <= c.kappa < 10.0


def test_3d_lo_both_names_and_jacobi():
    from ngsamg_amd import NgsAMG
    p = fem.poisson_fast((25, 25, 25), dirichlet="right|top")
    for name, kw, ms in (("NgsAMG.h1_scal", {}, 40), ("ngs_amg.h1_scal", {"ngs_amg_sm_type": "jacobi"}, 60),
                         ("NgsAMG.h1_scal", {"ngs_amg_mg_cycle": "W"}, 30), ("NgsAMG.h1_scal", {"ngs_amg_sm_symm": True}, 30)):
        c = NgsAMG.Preconditioner(_mat(p), name, freedofs=p.free, **kw)
        Solve(_mat(p), p.load, c, ms=ms, tol=1e-10)


@pytest.mark.parametrize("dim", [2, 3])
def test_vec_h1(dim):
    """h1_2d / h1_3d: vector-valued H1 (reference tests/h1/simple/test_vec.py): block-diagonal Laplacian"""
    from ngsamg_amd import NgsAMG
    shape = (41, 41) if dim == 2 else (15, 15, 15)
    p = fem.poisson_fast(shape, dirichlet="left|top")
    val = p.val[:, None, None] * np.eye(dim)[None]
    A = Matrix(p.n, p.n, dim, dim, p.rowptr, p.col, val)
    cls = NgsAMG.h1_2d if dim == 2 else NgsAMG.h1_3d
    c = cls(A, p.free, ngs_amg_max_coarse_size=10, ngs_amg_dim=dim)
    rhs = np.repeat(p.load, dim) * np.tile(np.arange(1, dim + 1), p.n)
    Solve(A, rhs, c, ms=50, tol=1e-10)
    assert c.GetBlockSize(0) == dim and c.GetBlockSize(1) == dim


@pytest.mark.parametrize("rot", [False, True])
def test_elast_3d_lo(rot):
    """reference tests/elasticity/mdim/simple/test_3d_lo.py (test_3d_lo, test_3d_lo_R): mu=1, lam=0, tol 1e-6, ms 40"""
    from ngsamg_amd import NgsAMG
    # test_3d_lo: beam 10 x 1 x 1; test_3d_lo_R (rotations): beam 2 x 1 x 1; both maxh = 0.25
    shape, ext = ((9, 5, 5), (2.0, 1.0, 1.0)) if rot else ((41, 5, 5), (10.0, 1.0, 1.0))
    p = fem.elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.0, rotations=rot, extent=ext)
    c = NgsAMG.elast_3d(_mat(p), p.free, coords=p.coords, ngs_amg_max_coarse_size=10)
    Solve(_mat(p), p.load, c, ms=40, tol=1e-6)           # the reference's budget
    assert c.GetBlockSize(0) == (6 if rot else 3) and c.GetBlockSize(1) == 6


@pytest.mark.parametrize("rot", [False, True])
def test_elast_3d_lo_edge_mats(rot):
    """the same two problems with ngs_amg_edge_mats: the matrix-valued smoothed prolongation of the reference (general 6x6 /
    3x6 blocks, no rigid-body storage) runs through the general block transfer kernels; the reference's budget holds and the
    application equals the oracle's on the same hierarchy"""
    from ngsamg_amd import NgsAMG
    from oracle.pyoracle import Oracle
    shape, ext = ((9, 5, 5), (2.0, 1.0, 1.0)) if rot else ((41, 5, 5), (10.0, 1.0, 1.0))
    p = fem.elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.0, rotations=rot, extent=ext)
    c = NgsAMG.elast_3d(_mat(p), p.free, coords=p.coords, ngs_amg_max_coarse_size=10, ngs_amg_edge_mats=True, ngs_amg_sm_type="jacobi")
    H = c.GetHierarchy()
    dev = c.GetAMGMatrix()._dev
    assert dev.matrix_info(0, "P")["fmt"] != "rigid-body"        # (a tiny coarse level may consist of piecewise rows Q(t) only)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    x = np.zeros_like(b)
    c.Mult(b, x)
    ref = Oracle(H.levels, sm_type="jacobi").apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    c2 = NgsAMG.elast_3d(_mat(p), p.free, coords=p.coords, ngs_amg_max_coarse_size=10, ngs_amg_edge_mats=True)
    Solve(_mat(p), p.load, c2, ms=40, tol=1e-6)           # the reference's budget (default smoother)


@pytest.mark.parametrize("rot", [False, True])
def test_elast_2d_lo(rot):
    """reference tests/elasticity/mdim/simple/test_2d_lo.py (test_2d_lo, test_2d_lo_R): beam 10 x 1, maxh = 0.1, mu=1, lam=0,
    max_coarse_size 20, ms 50; 2x2 displacement blocks (3x3 with the rotation) on level 0, 3x3 below"""
    from ngsamg_amd import NgsAMG
    from oracle.pyoracle import Oracle
    p = fem.elasticity_fast((101, 11), dirichlet="left", mu=1.0, lam=0.0, rotations=rot, extent=(10.0, 1.0))
    c = NgsAMG.elast_2d(_mat(p), p.free, coords=p.coords, ngs_amg_max_coarse_size=20, ngs_amg_rots=rot)
    Solve(_mat(p), p.load, c, ms=50, tol=1e-6)           # the reference's budget
    assert c.GetBlockSize(0) == (3 if rot else 2) and c.GetBlockSize(1) == 3
    # and the application itself against the oracle on the same hierarchy
    H = c.GetHierarchy()
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    x = np.zeros_like(b)
    c.Mult(b, x)
    # default smoother: Gauss-Seidel in the block-hybrid form where a level is large enough (block levels too since round 3);
    # the oracle runs the same blocks, colours and modified diagonals, multicolour order on the other levels
    from tests.hgs_oracle import hgs_levels
    lv, types = hgs_levels(H.levels, c.GetAMGMatrix()._dev.hgs)
    ref = Oracle(lv, sm_type=types).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("jump", [1e1, 1e2, 1e4, 1e6])
def test_elast_2d_material_jumps(jump, rot):
    """reference tests/elasticity/mdim/jump/test_2d_jump_lo.py: stiff inner squares (mu jump 1e1 ... 1e6), max_coarse_size 10,
    budget 50 -- met for every jump with the DEFAULT setup: the SPW agglomeration carries the scale of what a vertex has
    swallowed (maxTrOD, spw_agg_impl.hpp:600-626), so a collapsed rigid inclusion does not absorb its soft neighbours.
    (The target-driven pairwise rounds of the earlier builds, ngs_amg_spw=False, needed 49-128 iterations for jumps >= 1e4.)"""
    from ngsamg_amd import NgsAMG

    def coef(X):
        x, y = X[..., 0], X[..., 1]
        inner = ((np.abs(x - 0.3) < 0.1) | (np.abs(x - 0.7) < 0.1)) & ((np.abs(y - 0.3) < 0.1) | (np.abs(y - 0.7) < 0.1))
        return np.where(inner, jump, 1.0)

    p = fem.elasticity_fast((41, 41), dirichlet="left", mu=1.0, lam=0.0, rotations=rot, coef=coef)
    c = NgsAMG.elast_2d(_mat(p), p.free, coords=p.coords, ngs_amg_max_coarse_size=10, ngs_amg_rots=rot)
    sol, cg = Solve(_mat(p), p.load, c, ms=50, tol=1e-6)
    assert cg.iterations <= 40


def test_smoother_map_and_cinv_surface():
    from ngsamg_amd import NgsAMG
    from oracle.pyoracle import Oracle
    p = fem.poisson_fast((17, 17, 17))
    c = NgsAMG.h1_scal(_mat(p), p.free, ngs_amg_max_coarse_size=20)
    H = c.GetHierarchy()
    # default smoother: Gauss-Seidel in the block-hybrid form; the oracle runs the same blocks, colours and modified diagonal
    from tests.hgs_oracle import hgs_levels
    info = c.GetAMGMatrix()._dev.hgs
    assert info[0] is not None
    lv, types = hgs_levels(H.levels, info)
    orc = Oracle(lv, sm_type=types)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    sm = c.GetSmoother(0)
    x, xo = np.zeros(p.n), np.zeros(p.n)
    sm.SmoothK(2, x, b, None, False, True, True)
    ro = np.zeros(p.n)
    orc.smooth(0, xo, b, ro, False, True, True)
    orc.smooth(0, xo, b, ro, True, True, False)
    assert np.linalg.norm(x - xo) < 1e-10 * np.linalg.norm(xo)
    # DOFMap round trip and CINV
    m = c.GetMap()
    assert m.GetNLevels() == c.GetNLevels()
    xc = m.CreateVector(1)
    m.GetStep(0).TransferF2C(b, xc)
    assert np.linalg.norm(xc - orc.transfer_f2c(0, b)) < 1e-12 * np.linalg.norm(xc)
    sol = np.zeros(p.n)
    c.CINV(sol, b)
    # Galerkin coarse-grid correction: P_all A_L^-1 P_all^T b
    ref = b.copy()
    for l in range(H.n_levels - 1):
        ref = orc.transfer_f2c(l, ref)
    ref = orc.coarse_solve(ref)
    for l in range(H.n_levels - 2, -1, -1):
        up = np.zeros(H.levels[l].n)
        orc.add_c2f(l, 1.0, up, ref)
        ref = up
    assert np.linalg.norm(sol - ref) < 1e-9 * np.linalg.norm(ref)
    bf = c.GetBF(level=1, dof=3, comp=0)
    # a hat-like coarse basis function (rows smoothed with the level matrix may carry small negative weights where the matrix
    # has positive off-diagonal entries, as in the reference's classic branch)
    assert abs(bf.max() - 1.0) < 0.5 and bf.min() > -0.25 and bf.sum() > 0.0
    # stand-alone smoothers
    A = _mat(p)
    js = NgsAMG.CreateJacobiSmoother(A, p.free)
    x1 = np.zeros(p.n)
    js.Smooth(x1, b, None, False, False, True)
    assert np.allclose(x1, 0.9 * H.levels[0].dinv * b, rtol=1e-14, atol=1e-14)
    gs = NgsAMG.CreateHybridGSS(A, p.free)
    x2, x3 = np.zeros(p.n), np.zeros(p.n)
    gs.Smooth(x2, b)
    Oracle(H.levels[:1], sm_type="gs_mc", clev="none").smooth(0, x3, b, np.zeros(p.n), False, False, False)
    assert np.linalg.norm(x2 - x3) < 1e-10 * np.linalg.norm(x3)


def test_error_behaviour():
    from ngsamg_amd import NgsAMG
    p = fem.poisson_fast((9, 9))
    with pytest.raises(NgsAMGError):
        NgsAMG.Preconditioner(_mat(p), "ngs_amg.nonsense")
    c = NgsAMG.h1_scal()
    with pytest.raises(NgsAMGError):
        c.Mult(np.zeros(3), np.zeros(3))                    # not finalized (reference amg_pc.cpp:446)
    with pytest.raises(NgsAMGError):
        c.FinalizeLevel(None)                               # reference amg_pc.cpp:430
    e = fem.elasticity_fast((4, 4, 4))
    with pytest.raises(NgsAMGError):
        NgsAMG.elast_3d(_mat(e), e.free)                    # coordinates missing


def test_sub_amg_matrix_concatenated_steps_and_utils():
    """python_solve.cpp SubAMGMatrix, python_coarse.cpp ConcStep / Concatenate / ProjectMatrix, python_utils.cpp helpers"""
    from ngsamg_amd import NgsAMG
    from oracle.pyoracle import Oracle
    p = fem.poisson_fast((15, 14, 13))
    c = NgsAMG.h1_scal(_mat(p), p.free, ngs_amg_max_coarse_size=10, ngs_amg_sm_type="jacobi")
    amg = c.GetAMGMatrix()
    H = c.GetHierarchy()
    assert amg.GetNLevels() >= 3
    sub = amg.SubAMGMatrix(1)
    n1 = sub.height
    rng = np.random.default_rng(0)
    b1 = rng.standard_normal(n1) * H.levels[1].free
    x1 = np.zeros(n1)
    sub.Mult(b1, x1)
    ref = Oracle(H.levels[1:], sm_type="jacobi").apply(b1)
    assert np.linalg.norm(x1 - ref) <= 1e-12 * np.linalg.norm(ref)
    m = amg.GetMap()
    P02 = m.ConcStep(0, 2).to_scipy()
    ref02 = (H.levels[0].P.to_scipy() @ H.levels[1].P.to_scipy())
    assert abs(P02 - ref02).max() < 1e-14
    assert abs(m.GetStep(0).Concatenate(m.GetStep(1)).to_scipy() - ref02).max() < 1e-14
    A1 = m.GetStep(0).ProjectMatrix(H.levels[0].A).to_scipy()
    assert abs(A1 - H.levels[1].A.to_scipy()).max() <= 1e-12 * abs(A1).max()       # Galerkin identity
    assert m.SubMap(1).GetNLevels() == amg.GetNLevels() - 1
    # utils
    A = H.levels[0].A
    AA = NgsAMG.SparseMM(A, A).to_scipy()
    As = A.to_scipy()
    assert abs(AA - As @ As).max() <= 1e-12 * abs(AA).max()
    assert NgsAMG.GetMemoryUse(A) == A.rowptr.nbytes + A.col.nbytes + A.val.nbytes
    blocks = [np.arange(k, min(k + 5, A.n_rows)) for k in range(0, A.n_rows, 5)]
    R = NgsAMG.RestrictMatrixToBlocks(A, blocks).to_scipy().tocoo()
    assert np.all(R.row // 5 == R.col // 5) and R.nnz > 0
    Z = As.copy()
    Z.data[::3] = 0.0
    Zc = NgsAMG.CompressSparseMatrix(Z.tocsr())
    assert Zc.nnz == np.count_nonzero(Z.data) and abs(Zc.to_scipy() - Z).max() == 0.0
    assert NgsAMG.ToSparseMatrix(Z.tocsr(), compress=True).nnz == Zc.nnz


def test_direct_inverse_smoother():
    """CreateHybridDISmoother on one rank = Richardson with the exact inverse: one step solves the system"""
    from ngsamg_amd import NgsAMG
    p = fem.poisson_fast((9, 8, 7))
    sm = NgsAMG.CreateHybridDISmoother(_mat(p), p.free)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(p.n) * p.free
    x = np.zeros(p.n)
    res = np.zeros(p.n)
    sm.Smooth(x, b, res, False, True, False)
    A = _mat(p).to_scipy()
    assert np.linalg.norm((b - A @ x) * p.free) <= 1e-10 * np.linalg.norm(b)
    assert np.linalg.norm((res - (b - A @ x)) * p.free) <= 1e-10 * np.linalg.norm(b)


@pytest.mark.parametrize("jump", [1e1, 1e2, 1e4, 1e6])
@pytest.mark.parametrize("geom", ["squares", "fibers"])
def test_2d_coefficient_jumps(jump, geom):
    """reference tests/h1/jump/test_2d_jump_lo.py: coefficient jumps 1e1 ... 1e6 between an outer material and inner
    squares / horizontal fibres, max_coarse_size 5, tol 1e-6, budgets 25-40 iterations.
    Measured with this build (tools/jump_probe.py): inner squares 18-19 (GS) / 14 (block GS) iterations for every jump;
    fibres 22 / 32 / 60 / 81 with point GS (the build's simple aggregation + prolongation smoothing is not robust for
    strongly anisotropic coefficient patterns: host setup quality) but 15 / 14 / 13 / 11 with the
    block smoother, inside the reference's budget."""
    from ngsamg_amd import ngs_amg

    def coef(X):
        x, y = X[..., 0], X[..., 1]
        if geom == "squares":          # four inner squares of material b
            inner = ((np.abs(x - 0.3) < 0.1) | (np.abs(x - 0.7) < 0.1)) & ((np.abs(y - 0.3) < 0.1) | (np.abs(y - 0.7) < 0.1))
        else:                          # horizontal fibres
            inner = (np.floor(y * 10) % 2 == 1) & (np.abs(x - 0.5) < 0.4)
        return np.where(inner, jump, 1.0)

    diri = "left|right|top|bottom" if geom == "squares" else "top|bottom"
    p = fem.poisson_fast((81, 81), dirichlet=diri, coef=coef)
    # reference budgets: 25 (squares: default smoother; fibres: bgs).  Point GS on the fibres is not a reference test (its
    # fibre test sets sm_type = bgs); with the SPW agglomeration (default) it stays inside the same budget
    budget = {"gs": 25, "bgs": 25}
    for sm in ("gs", "bgs"):
        c = ngs_amg.Preconditioner(_mat(p), "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2,
                                   ngs_amg_sm_type=sm)
        Solve(_mat(p), p.load, c, ms=budget[sm], tol=1e-6)
    # a weaker prolongation smoothing (flag ngs_amg_sp_omega, reference default 1.0, its example uses 0.8) keeps point GS
    # inside the reference's budget on the fibres too: 23-27 iterations for every jump
    c = ngs_amg.Preconditioner(_mat(p), "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2,
                               ngs_amg_sm_type="gs", ngs_amg_sp_omega=0.5)
    Solve(_mat(p), p.load, c, ms=35, tol=1e-6)
    # the earlier builds' pairwise rounds (ngs_amg_spw=False) with their private vertex scales (ngs_amg_robust_soc): same budget
    c = ngs_amg.Preconditioner(_mat(p), "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2,
                               ngs_amg_sm_type="gs", ngs_amg_spw=False, ngs_amg_robust_soc=True)
    Solve(_mat(p), p.load, c, ms=25, tol=1e-6)
