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
