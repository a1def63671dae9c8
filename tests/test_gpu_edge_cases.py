"""Edge cases of the single-GPU apply path: degenerate hierarchies and inputs (one level, almost no free dofs, zero right-hand
side, thin domains), each against the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(p, H, sm, osm=None, cycle="V"):
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    dev = DeviceAMGMatrix(H, sm_type=sm, mg_cycle=cycle, device=0)
    if sm == "hgs":
        from tests.hgs_oracle import hgs_levels
        lv, types = hgs_levels(H.levels, dev.hgs)
        orc = Oracle(lv, sm_type=types, cycle=cycle)
    else:
        orc = Oracle(H.levels, sm_type=osm or sm, cycle=cycle)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    x = dev.apply(b)
    ref = orc.apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * max(np.linalg.norm(ref), 1e-300)
    z = dev.apply(np.zeros_like(b))
    assert np.array_equal(z, np.zeros_like(b))
    return dev, x


@pytest.mark.parametrize("sm,osm", [("jacobi", None), ("gs", "gs_mc"), ("hgs", None)])
def test_single_level_hierarchy_is_the_exact_solve(sm, osm):
    """max_levels = 1: the cycle is the coarse solve alone (amg_matrix.cpp:217-247 on level 0)"""
    from ngsamg_amd import fem
    from ngsamg_amd.hierarchy import Hierarchy
    from tests.problems import to_matrix
    p = fem.poisson_fast((5, 4, 3))
    H = Hierarchy(to_matrix(p), p.free, p.coords, dim=3, energy=0, max_levels=1)
    assert H.n_levels == 1 and H.coarse_n == p.n
    dev, x = _run(p, H, sm, osm)
    A = to_matrix(p).to_scipy().toarray()
    f = p.free.astype(bool)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    exact = np.zeros(p.n)
    exact[f] = np.linalg.solve(A[np.ix_(f, f)], b[f])
    assert np.allclose(x, exact, rtol=0, atol=1e-11 * np.abs(exact).max())


@pytest.mark.parametrize("sm,osm", [("jacobi", None), ("gs", "gs_mc"), ("hgs", None)])
@pytest.mark.parametrize("shape", [(300, 3), (2, 2, 400), (65, 64)])
def test_thin_and_odd_domains(sm, osm, shape):
    """chains and slabs: aggregates degenerate, rows are short, slices are ragged"""
    from tests.problems import poisson_case
    p, H = poisson_case(shape, "right|top", 8)
    _run(p, H, sm, osm)
    _run(p, H, sm, osm, cycle="W")


@pytest.mark.parametrize("sm,osm", [("jacobi", None), ("gs", "gs_mc"), ("hgs", None)])
def test_almost_everything_dirichlet(sm, osm):
    """only a small patch of free dofs in a large masked matrix: non-free rows must stay untouched (0) on every level"""
    from ngsamg_amd import fem
    from ngsamg_amd.hierarchy import Hierarchy
    from tests.problems import to_matrix
    p = fem.poisson_fast((24, 24, 24))
    free = np.zeros(p.n, dtype=np.uint8)
    idx = np.arange(p.n).reshape(24, 24, 24)
    free[idx[8:15, 9:14, 10:16].ravel()] = 1
    p.free = free
    H = Hierarchy(to_matrix(p), free, p.coords, dim=3, energy=0, max_coarse_size=10)
    dev, x = _run(p, H, sm, osm)
    assert np.all(x[free == 0] == 0.0)


@pytest.mark.parametrize("kind,sm", [("poisson", "jacobi"), ("poisson", "gs"), ("elast6", "jacobi")])
def test_large_coarsest_level_is_inverted_on_the_device(kind, sm, monkeypatch):
    """The reference always inverts the coarsest matrix (BaseAMGPC::CoarseLevelInv, amg_pc.cpp:843-928).  When the host
    setup declines (more than NGSAMG_HOST_COARSE_MAX = 4096 unknowns) amgx_create forms the dense inverse on the device
    (dense_spd.hpp: blocked Gauss-Jordan, f64 MFMA trailing updates): same cycle result as with the oracle's exact coarse
    solve.  The limit is lowered here so that the oracle's dense Cholesky stays fast."""
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("NGSAMG_HOST_COARSE_MAX", "64")
    if kind == "poisson":
        p = fem.poisson_fast((26, 25, 24), dirichlet="right|top")
        H = Hierarchy(Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val), p.free, p.coords, dim=3, energy=0, max_levels=2)
    else:
        p = fem.elasticity_fast((9, 8, 8), dirichlet="left", mu=1.0, lam=0.5, rotations=True)
        H = Hierarchy(Matrix(p.n, p.n, 6, 6, p.rowptr, p.col, p.val), p.free, p.coords, dim=3, energy=1, max_levels=2, regularize_cmats=0)
    assert H.n_levels == 2 and H.coarse_n == 0
    nc = H.levels[-1].n * H.levels[-1].bs
    assert nc > 128 and nc % 64 != 0            # several tiles, with padding
    dev = DeviceAMGMatrix(H, sm_type=sm, device=0)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    x = np.empty_like(b)
    dev.Mult(b, x)
    ref = Oracle(H.levels, sm_type="jacobi" if sm == "jacobi" else "gs_mc").apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    # the stand-alone coarse solve: A_c x = r on the free dofs
    r = rng.standard_normal(nc) * np.repeat(H.levels[-1].free, H.levels[-1].bs)
    xc = np.empty(nc)
    dev.CoarseSolve(r, xc)
    Ac = H.levels[-1].A.to_scipy()
    f = np.repeat(H.levels[-1].free.astype(bool), H.levels[-1].bs)
    assert np.linalg.norm((Ac @ xc - r)[f]) <= 1e-9 * np.linalg.norm(r) and np.all(xc[~f] == 0.0)
