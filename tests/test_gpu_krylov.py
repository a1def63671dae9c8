"""Device-resident Krylov solvers behind the C ABI (amgx_pcg, amgx_gmres: hand-written BLAS-1 kernels) against the oracle's
PCG (same recurrence, same stopping rule as NGSolve's CGSolver in the reference's drivers, tests/h1/amg_utils.py:337-363)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(shape, sm):
    from tests.problems import poisson_case
    from ngsamg_amd.device import DeviceAMGMatrix
    p, H = poisson_case(shape, "right|top", 20)
    return p, H, DeviceAMGMatrix(H, sm_type=sm, device=0)


@pytest.mark.parametrize("sm,osm", [("jacobi", "jacobi"), ("gs", "gs_mc")])
@pytest.mark.parametrize("device_vectors", [True, False])
def test_native_pcg_matches_oracle_pcg(sm, osm, device_vectors):
    import torch
    from ngsamg_amd.krylov import NativeCGSolver
    from oracle.pyoracle import Oracle
    p, H, dev = _case((25, 25, 25), sm)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    cg = NativeCGSolver(dev, dev, tol=1e-10, maxsteps=100)
    x = cg.Solve(torch.from_numpy(b).cuda() if device_vectors else b)
    x = x.cpu().numpy() if device_vectors else x
    xo, it, errs = Oracle(H.levels, sm_type=osm).pcg(b, tol=1e-10, maxit=100)
    assert abs(cg.iterations - it) <= 1
    k = min(cg.iterations, it)
    assert np.allclose(cg.errors[:k], errs[:k], rtol=1e-6)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
    A = H.levels[0].A.to_scipy()
    f = p.free.astype(bool)
    assert np.linalg.norm((A @ x - b)[f]) <= 1e-8 * np.linalg.norm(b)


def test_native_cg_without_preconditioner():
    """use_precond = 0: plain CG on the stored level-0 matrix (which keeps its Dirichlet rows, i.e. is singular: only the
    recurrence is compared, iteration by iteration, with the oracle's)"""
    from ngsamg_amd.krylov import NativeCGSolver
    from oracle.pyoracle import Oracle
    p, H, dev = _case((13, 13, 13), "jacobi")
    rng = np.random.default_rng(1)
    b = rng.standard_normal(p.n) * p.free
    cg = NativeCGSolver(dev, None, tol=1e-30, maxsteps=15)
    cg.Solve(b)
    _, it, errs = Oracle(H.levels, sm_type="jacobi").pcg(b, tol=1e-30, maxit=15, precond=False)
    assert cg.iterations == it == 15
    assert np.allclose(cg.errors, errs, rtol=1e-8)


@pytest.mark.parametrize("restart", [5, 30])
def test_native_gmres_converges_like_pcg(restart):
    import torch
    from ngsamg_amd.krylov import NativeCGSolver, NativeGMResSolver
    p, H, dev = _case((25, 25, 25), "jacobi")
    rng = np.random.default_rng(2)
    b = rng.standard_normal(p.n) * p.free
    bd = torch.from_numpy(b).cuda()
    gm = NativeGMResSolver(dev, dev, tol=1e-10, maxsteps=200, restart=restart)
    x = gm.Solve(bd).cpu().numpy()
    cg = NativeCGSolver(dev, dev, tol=1e-10, maxsteps=200)
    xc = cg.Solve(bd).cpu().numpy()
    assert gm.errors[-1] <= 1e-10 * gm.errors[0]
    assert all(e2 <= e1 * (1 + 1e-12) for e1, e2 in zip(gm.errors[:restart], gm.errors[1:restart + 1]))     # monotone inside a cycle
    # full GMRES minimises the preconditioned residual over the same Krylov space CG works in: no more iterations than CG
    if restart >= 30:
        assert gm.iterations <= cg.iterations + 2
    A = H.levels[0].A.to_scipy()
    f = p.free.astype(bool)
    assert np.linalg.norm((A @ x - b)[f]) <= 1e-7 * np.linalg.norm(b)
    assert np.linalg.norm(x - xc) <= 1e-6 * np.linalg.norm(xc)


@pytest.mark.parametrize("restart,sm", [(5, "jacobi"), (30, "jacobi"), (12, "gs")])
def test_native_gmres_history_equals_oracle_gmres(restart, sm):
    """amgx_gmres against the oracle's GMRES (plain C, modified Gram-Schmidt, written independently): same iteration count,
    error history equal to 1e-6 of the initial error, same solution"""
    import torch
    from ngsamg_amd.krylov import NativeGMResSolver
    from oracle.pyoracle import Oracle
    p, H, dev = _case((25, 25, 25), sm)
    rng = np.random.default_rng(4)
    b = rng.standard_normal(p.n) * p.free
    gm = NativeGMResSolver(dev, dev, tol=1e-9, maxsteps=150, restart=restart)
    x = gm.Solve(torch.from_numpy(b).cuda()).cpu().numpy()
    xo, ito, erro = Oracle(H.levels, sm_type="jacobi" if sm == "jacobi" else "gs_mc").gmres(b, tol=1e-9, maxit=150, restart=restart)
    assert gm.iterations == ito
    errs = np.asarray(gm.errors)
    assert errs.shape == erro.shape and np.all(np.abs(errs - erro) <= 1e-6 * erro[0])
    assert np.allclose(errs[:4], erro[:4], rtol=1e-9)
    assert np.linalg.norm(x - xo) <= 1e-7 * np.linalg.norm(xo)


@pytest.mark.parametrize("sm,osm", [("jacobi", "jacobi"), ("gs", "gs_mc")])
def test_single_reduction_pcg_history_equals_classical(sm, osm):
    """AMGX_PCG_SINGLE_REDUCTION (Chronopoulos / Gear form: one reduction point, three launches per iteration) against the oracle's
    classical PCG: same iteration count (+-1), histories equal to 1e-6, same solution"""
    import torch
    from ngsamg_amd.krylov import NativeCGSolver
    from oracle.pyoracle import Oracle
    p, H, dev = _case((25, 25, 25), sm)
    rng = np.random.default_rng(7)
    b = rng.standard_normal(p.n) * p.free
    cg = NativeCGSolver(dev, dev, tol=1e-10, maxsteps=100, single_reduction=True)
    x = cg.Solve(torch.from_numpy(b).cuda()).cpu().numpy()
    xo, it, errs = Oracle(H.levels, sm_type=osm).pcg(b, tol=1e-10, maxit=100)
    assert abs(cg.iterations - it) <= 1
    k = min(cg.iterations, it)
    assert np.allclose(cg.errors[:k], errs[:k], rtol=1e-6)
    assert np.linalg.norm(x - xo) <= 1e-8 * np.linalg.norm(xo)
