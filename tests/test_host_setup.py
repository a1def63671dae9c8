"""Host setup (cold path): generators and hierarchy invariants the reference itself relies on
(P^T exact transpose with sorted columns, Galerkin A_c = P^T A P, prolongation preserves
constants / rigid body modes)."""
import numpy as np
import pytest
import scipy.sparse as sp

from ngsamg_amd import fem
from ngsamg_amd._lib import Matrix
from tests.problems import poisson_case, elasticity_case, to_matrix


@pytest.mark.parametrize("shape", [(9, 9), (6, 7, 8)])
def test_poisson_fast_matches_numpy(shape):
    a, b = fem.poisson(shape), fem.poisson_fast(shape)
    assert np.array_equal(a.rowptr, b.rowptr) and np.array_equal(a.col, b.col)
    assert np.abs(a.val - b.val).max() < 1e-14
    assert np.abs(a.load - b.load).max() < 1e-15
    A = b.to_scipy()
    assert abs(A - A.T).max() < 1e-14
    assert np.abs(A.sum(axis=1)).max() < 1e-13          # constants in the kernel (Neumann operator)
    assert np.all(np.diff(b.col.astype(np.int64))[np.setdiff1d(np.arange(b.nnz - 1), b.rowptr[1:-1] - 1)] > 0)


@pytest.mark.parametrize("shape,rot", [((5, 6), False), ((5, 6), True), ((4, 5, 6), False), ((4, 5, 6), True)])
def test_elasticity_fast_matches_numpy_and_kernel(shape, rot):
    a = fem.elasticity(shape, rotations=rot, lam=0.7, mu=1.3)
    b = fem.elasticity_fast(shape, rotations=rot, lam=0.7, mu=1.3)
    assert np.array_equal(a.col, b.col)
    assert np.abs(a.val - b.val).max() < 1e-13
    E = b.to_scipy()
    assert abs(E - E.T).max() < 1e-13
    dim = len(shape)
    X = b.coords
    if dim == 3:
        w = np.array([0.3, -0.2, 0.5])
        u = np.cross(w, X)
        full = np.concatenate([u, np.tile(w, (b.n, 1))], axis=1) if rot else u
        assert np.abs(E @ full.ravel()).max() < 1e-12      # rigid rotation has zero energy
    t = np.zeros((b.n, b.bs))
    t[:, 0] = 1.0
    assert np.abs(E @ t.ravel()).max() < 1e-12             # translation


@pytest.mark.parametrize("shape,diri,mcs", [((33, 33), "left|top", 5), ((17, 17, 17), "right|top", 20)])
def test_h1_hierarchy_invariants(shape, diri, mcs):
    p, H = poisson_case(shape, diri, mcs)
    assert H.n_levels >= 3
    for l, L in enumerate(H.levels[:-1]):
        P, PT, A = L.P.to_scipy(), L.PT.to_scipy(), L.A.to_scipy()
        Ac = H.levels[l + 1].A.to_scipy()
        assert abs(PT - P.T).max() == 0.0
        assert abs(Ac - P.T @ A @ P).max() < 1e-12 * abs(A).max() * 50
        for M in (L.P, L.PT, L.A):       # columns strictly ascending per row
            for i in range(min(M.n_rows, 200)):
                c = M.col[M.rowptr[i]:M.rowptr[i + 1]]
                assert np.all(np.diff(c) > 0)
        rs = np.asarray(P.sum(axis=1)).ravel()
        free = L.free.astype(bool)
        assert np.allclose(rs[free], 1.0, atol=1e-14)      # constants are reproduced on free vertices
        assert np.all(rs[~free] == 0.0)                    # Dirichlet vertices are not in the coarse space
        assert len(np.unique(L.P.col)) == L.P.n_cols       # no empty coarse column
        assert np.diff(L.P.rowptr).max() <= 3              # sp_max_per_row
        # smoother diagonal
        d = A.diagonal()
        assert np.allclose(L.dinv[free], 1.0 / d[free])
        assert np.all(L.dinv[~free] == 0.0)
        # colouring is proper
        col = L.color
        C = sp.csr_matrix(A)
        C.setdiag(0)
        C.eliminate_zeros()
        i, j = C.nonzero()
        m = (col[i] >= 0) & (col[j] >= 0)
        assert np.all(col[i][m] != col[j][m])
        assert np.all((col >= 0) == free)
    Lc = H.levels[-1]
    ci = H.coarse_inv.reshape(H.coarse_n, H.coarse_n)
    assert np.abs(ci @ Lc.A.to_scipy().toarray() - np.eye(H.coarse_n)).max() < 1e-9
    assert H.levels[-1].n <= mcs or H.n_levels == 10
    assert 1.0 < H.operator_complexity() < 2.0


@pytest.mark.parametrize("rot", [False, True])
def test_elasticity_hierarchy_preserves_rigid_body_modes(rot):
    p, H = elasticity_case((7, 6, 5), rot, 10)
    assert H.levels[0].bs == (6 if rot else 3)
    assert all(L.bs == 6 for L in H.levels[1:])
    L0, L1 = H.levels[0], H.levels[1]
    P = L0.P.to_scipy()
    assert (L0.P.br, L0.P.bc) == ((6, 6) if rot else (3, 6))
    rng = np.random.default_rng(2)
    t, w = rng.standard_normal(3), rng.standard_normal(3)
    xc, xf = L1.coords, L0.coords
    coarse = np.concatenate([t + np.cross(w, xc), np.tile(w, (L1.n, 1))], axis=1).ravel()
    fine_u = t + np.cross(w, xf)
    fine = (np.concatenate([fine_u, np.tile(w, (L0.n, 1))], axis=1) if rot else fine_u)
    got = (P @ coarse).reshape(L0.n, -1)
    free = L0.free.astype(bool)
    assert np.abs(got[free] - fine[free]).max() < 1e-12
    # Galerkin + symmetric coarse operator, zero energy of rigid body modes on the (unclamped part of the) coarse level
    A1 = L1.A.to_scipy()
    assert abs(A1 - A1.T).max() < 1e-10 * abs(A1).max()
    assert abs(A1 - P.T @ L0.A.to_scipy() @ P).max() < 1e-10 * abs(A1).max()


def test_transpose_and_matmul_helpers():
    import ctypes as C
    from ngsamg_amd import _lib
    lib = _lib.host()
    rng = np.random.default_rng(0)
    A = sp.random(40, 30, density=0.2, random_state=1, format="csr")
    B = sp.random(30, 25, density=0.2, random_state=2, format="csr")
    MA, MB = Matrix.from_scipy(A), Matrix.from_scipy(B)
    da, db = MA.desc(), MB.desc()
    rp = np.zeros(41, dtype=np.int64)
    _lib.hcheck(lib.amgh_matmul(C.byref(da), C.byref(db), _lib.ptr(rp, C.c_int64), None, None))
    col = np.zeros(rp[-1], dtype=np.int32)
    val = np.zeros(rp[-1])
    _lib.hcheck(lib.amgh_matmul(C.byref(da), C.byref(db), _lib.ptr(rp, C.c_int64), _lib.ptr(col, C.c_int32), _lib.ptr(val, C.c_double)))
    Cm = sp.csr_matrix((val, col, rp), shape=(40, 25))
    assert abs(Cm - A @ B).max() < 1e-14
    rt = np.zeros(31, dtype=np.int64)
    _lib.hcheck(lib.amgh_transpose_count(C.byref(da), _lib.ptr(rt, C.c_int64)))
    ct = np.zeros(rt[-1], dtype=np.int32)
    vt = np.zeros(rt[-1])
    _lib.hcheck(lib.amgh_transpose_fill(C.byref(da), _lib.ptr(rt, C.c_int64), _lib.ptr(ct, C.c_int32), _lib.ptr(vt, C.c_double)))
    assert abs(sp.csr_matrix((vt, ct, rt), shape=(30, 40)) - A.T).max() == 0.0


def test_setup_errors():
    from ngsamg_amd._lib import NgsAMGError
    from ngsamg_amd.hierarchy import Hierarchy
    p = fem.poisson_fast((5, 5))
    A = to_matrix(p)
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free[:-1], p.coords, dim=2)
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, p.coords, dim=4)
    e = fem.elasticity_fast((4, 4, 4))
    with pytest.raises(NgsAMGError):
        Hierarchy(to_matrix(e), e.free, None, dim=3, energy=1)     # elasticity needs coordinates


@pytest.mark.parametrize("shape,dim", [((65, 65), 2), ((21, 21, 21), 3)])
def test_multistep_concatenated_prolongation(shape, dim):
    """ngs_amg_enable_multistep (reference h1_impl.hpp:331): a level is reached by several concatenated coarsening steps;
    the concatenated P still reproduces constants, P^T is its exact transpose and the coarse matrix is the Galerkin product"""
    from ngsamg_amd.hierarchy import Hierarchy
    p = fem.poisson_fast(shape, dirichlet="left|top")
    H = Hierarchy(to_matrix(p), p.free, p.coords, dim=dim, energy=0, max_coarse_size=5, enable_multistep=1)
    assert H.n_levels >= 3
    for l, L in enumerate(H.levels[:-1]):
        P, PT, A = L.P.to_scipy(), L.PT.to_scipy(), L.A.to_scipy()
        Ac = H.levels[l + 1].A.to_scipy()
        assert abs(PT - P.T).max() == 0.0
        assert abs(Ac - P.T @ A @ P).max() < 1e-11 * abs(A).max() * 50
        rs = np.asarray(P.sum(axis=1)).ravel()
        free = L.free.astype(bool)
        assert np.allclose(rs[free], 1.0, atol=1e-13)
        assert L.agg is not None and L.agg.max() < H.levels[l + 1].n


def test_robust_soc_keeps_material_jumps_h_and_jump_independent():
    """ngs_amg_robust_soc: every vertex carries the largest edge weight collapsed inside it; a connection that is negligible on
    that scale is not a viable pairing, so stiff inclusions / fibres that have become single vertices do not absorb their
    soft surroundings (reference: accumulated vertex weights in the strength of connection of the SPW agglomerator)"""
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle

    def its_for(p, dim, energy, rs, **kw):
        A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
        H = Hierarchy(A, p.free, p.coords, dim=dim, energy=energy, robust_soc=rs, **kw)
        return Oracle(H.levels, sm_type="gs").pcg(p.load, tol=1e-6, maxit=200)[1], H

    for jump in (1e2, 1e6):
        def fibres(X):
            x, y = X[..., 0], X[..., 1]
            return np.where((np.floor(y * 10) % 2 == 1) & (np.abs(x - 0.5) < 0.4), jump, 1.0)

        def squares(X):
            x, y = X[..., 0], X[..., 1]
            inner = ((np.abs(x - 0.3) < 0.1) | (np.abs(x - 0.7) < 0.1)) & ((np.abs(y - 0.3) < 0.1) | (np.abs(y - 0.7) < 0.1))
            return np.where(inner, jump, 1.0)

        p = fem.poisson_fast((81, 81), dirichlet="top|bottom", coef=fibres)
        it1, H1 = its_for(p, 2, 0, 1, max_coarse_size=5)
        it0, _ = its_for(p, 2, 0, 0, max_coarse_size=5)
        assert it1 <= 20 and it1 < it0 and H1.coarse_n <= 64
        for rot in (False, True):
            e = fem.elasticity_fast((41, 41), dirichlet="left", mu=1.0, lam=0.0, rotations=rot, coef=squares)
            it1, H1 = its_for(e, 2, 1, 1, max_coarse_size=10, regularize_cmats=0 if rot else 1)
            it0, _ = its_for(e, 2, 1, 0, max_coarse_size=10, regularize_cmats=0 if rot else 1)
            assert it1 <= 20 and it1 < it0
    # a uniform problem is hardly affected
    p = fem.poisson_fast((33, 33, 33))
    it1, _ = its_for(p, 3, 0, 1, max_coarse_size=20)
    it0, _ = its_for(p, 3, 0, 0, max_coarse_size=20)
    assert abs(it1 - it0) <= 2
