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
