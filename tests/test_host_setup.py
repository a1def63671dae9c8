"""Host setup (cold path): generators and hierarchy invariants the reference itself relies on
(P^T exact transpose with sorted columns, Galerkin A_c = P^T A P, prolongation preserves
constants / rigid body modes)."""
import numpy as np
import pytest
import scipy.sparse as sp

from ngsamg_amd import fem
from ngsamg_amd._lib import Matrix, NgsAMGError
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
        # constants are reproduced wherever they are in the kernel of the row: rows smoothed with the level matrix ("classic"
        # branch of the reference's semi-aux prolongation) sum to 1 - omega * rowsum(A)_i / a_ii, all others to 1
        ra = np.asarray(A.sum(axis=1)).ravel() / A.diagonal()
        assert np.all(np.isclose(rs[free], 1.0, atol=1e-13) | np.isclose(rs[free], 1.0 - ra[free], atol=1e-12))
        assert np.allclose(rs[free & (np.abs(ra) < 1e-13)], 1.0, atol=1e-13)
        if l == 0:
            assert np.allclose(rs[free], 1.0, atol=1e-13)  # (level 0: rows next to Dirichlet vertices take the aux branch)
        assert np.all(rs[~free] == 0.0)                    # Dirichlet vertices are not in the coarse space
        assert len(np.unique(L.P.col)) == L.P.n_cols       # no empty coarse column
        assert np.diff(L.P.rowptr).max() <= 5              # sp_max_per_row_classic (3 = sp_max_per_row on the aux rows)
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
    # (aux_smoothed: every row of every factor sums to 1, so the concatenation reproduces constants exactly)
    H = Hierarchy(to_matrix(p), p.free, p.coords, dim=dim, energy=0, max_coarse_size=5, enable_multistep=1, prol_type="aux_smoothed")
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


def test_spw_agglomeration_is_robust_for_material_jumps():
    """The reference's SPW agglomeration (default, amgh_options.spw; spw_agg_impl.hpp): maxTrOD of a merged vertex keeps the
    scale of everything collapsed inside it (SPWAggData::Map, :600-626), so a stiff inclusion / fibre that has become one
    vertex does not absorb its soft surroundings: iteration counts stay bounded for every jump, where the target-driven
    pairwise rounds of the earlier builds (spw = 0, plain max of the live edges) degrade.  The reference's own tests for
    this: tests/elasticity/mdim/jump/test_2d_jump_lo.py, tests/h1/ jump cases (budgets 50)."""
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle

    def its_for(p, dim, energy, spw, **kw):
        A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
        H = Hierarchy(A, p.free, p.coords, dim=dim, energy=energy, spw=spw, **kw)
        return Oracle(H.levels, sm_type="gs").pcg(p.load, tol=1e-6, maxit=200)[1], H

    worst_old = 0
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
        assert it1 <= 20 and H1.coarse_n <= 64
        worst_old = max(worst_old, it0)
        for rot in (False, True):
            e = fem.elasticity_fast((41, 41), dirichlet="left", mu=1.0, lam=0.0, rotations=rot, coef=squares)
            it1, H1 = its_for(e, 2, 1, 1, max_coarse_size=10, regularize_cmats=0 if rot else 1)
            it0, _ = its_for(e, 2, 1, 0, max_coarse_size=10, regularize_cmats=0 if rot else 1)
            assert it1 <= 35                  # (the reference's budget for these cases is 50)
            worst_old = max(worst_old, it0)
    assert worst_old > 40                     # what the rule is there for
    # the private option of the earlier builds (vertex scales on top of the pairwise rounds) still does its job
    p = fem.poisson_fast((81, 81), dirichlet="top|bottom", coef=lambda X: np.where((np.floor(X[..., 1] * 10) % 2 == 1) & (np.abs(X[..., 0] - 0.5) < 0.4), 1e6, 1.0))
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, dim=2, energy=0, spw=0, robust_soc=1, max_coarse_size=5)
    assert Oracle(H.levels, sm_type="gs").pcg(p.load, tol=1e-6, maxit=200)[1] <= 20
    # ... and is refused next to the SPW agglomerator, which would silently ignore the vertex scales
    from ngsamg_amd._lib import NgsAMGError
    with pytest.raises(NgsAMGError, match="robust_soc needs spw = 0"):
        Hierarchy(A, p.free, p.coords, dim=2, energy=0, robust_soc=1, max_coarse_size=5)


def test_spw_pairing_rule_properties():
    """aggregate_spw restates the reference's pairing rule (FindNeib3Step, spw_agg_impl.hpp:637-775): spw_rounds pairing
    rounds => aggregates of at most 2^rounds vertices (+ orphans that joined them, JoiningIteration :1265-1365); a pair is
    only formed over a connection with soc >= 0.25 of the vertex's strongest; spw_rounds and the orphan round are options
    (spw_agg.hpp:28-32); the reference's 2D Poisson budget (tests/h1/test_2d_poisson.py: < 30 iterations at tol 1e-12 with
    Gauss-Seidel) holds at the size of BASELINE.json's configs[0] (224^2)"""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    p = fem.poisson_fast((40, 40), dirichlet="left|top")
    for rounds, orphan in ((1, 0), (2, 1), (3, 1), (3, 0)):
        H = Hierarchy(to_matrix(p), p.free, p.coords, dim=2, energy=0, max_levels=2, spw_rounds=rounds, spw_orphan_treatment=orphan)
        agg = H.levels[0].agg
        sizes = np.bincount(agg[agg >= 0])
        assert np.all(agg[~p.free.astype(bool)] == -1) and np.all(agg[p.free.astype(bool)] >= 0)
        # pairs of pairs: 2^rounds; an orphan round can add single vertices to a real aggregate
        assert sizes.max() <= 2 ** rounds + (6 if orphan else 0)
        if not orphan:
            assert sizes.max() <= 2 ** rounds
        if rounds == 3:
            assert 5.0 <= sizes.mean() <= 9.0
            if orphan:
                assert (sizes == 1).sum() <= (np.bincount(Hierarchy(to_matrix(p), p.free, p.coords, dim=2, energy=0, max_levels=2, spw_rounds=3,
                                                                    spw_orphan_treatment=0).levels[0].agg.clip(0)) == 1).sum()
    # a chain with one weak link: the weak link is never inside an aggregate
    import scipy.sparse as sp
    from ngsamg_amd._lib import Matrix
    n = 16
    w = np.ones(n - 1)
    w[7] = 1e-3
    L = sp.diags([-w, -w], [-1, 1]).tolil()
    L.setdiag(np.asarray(-L.sum(axis=1)).ravel() + 1e-3)
    Lc = sp.csr_matrix(L)
    Lc.sort_indices()
    A = Matrix(n, n, 1, 1, Lc.indptr.astype(np.int64), Lc.indices.astype(np.int32), Lc.data.copy())
    H = Hierarchy(A, np.ones(n, dtype=np.uint8), np.stack([np.arange(n) / n, np.zeros(n)], axis=1), dim=2, energy=0, max_levels=2, max_coarse_size=1)
    agg = H.levels[0].agg
    assert agg[7] != agg[8]
    # the reference's budget at the size of configs[0]
    p = fem.poisson_fast((224, 224), dirichlet="left|top")
    H = Hierarchy(to_matrix(p), p.free, p.coords, dim=2, energy=0, max_coarse_size=20)
    rng = np.random.default_rng(0)
    it = Oracle(H.levels, sm_type="gs", threads=8).pcg(rng.standard_normal(p.n) * p.free, tol=1e-12, maxit=100)[1]
    assert it < 30


def test_compact_sweep_blocks_colouring_and_l1_block_diagonal():
    """amgh_compact_blocks / amgh_coloring_blockids / amgh_hybrid_dinv_block_ids (block-hybrid Gauss-Seidel on block levels):
    blocks of at most max_rows block rows cover every row once and are far more compact than runs of consecutive rows; the
    colouring separates coupled rows of one block; the diagonal follows hybrid_smoother_utils.hpp:86-141"""
    import ctypes as C
    from ngsamg_amd import _lib
    lib = _lib.host()
    p = fem.elasticity_fast((8, 8, 40), dirichlet="left", mu=1.0, lam=0.5, rotations=True)
    A = to_matrix(p)
    d = A.desc()
    fr = np.ascontiguousarray(p.free, dtype=np.uint8)
    blk = np.zeros(p.n, dtype=np.int32)
    nb = C.c_int64()
    _lib.hcheck(lib.amgh_compact_blocks(C.byref(d), _lib.ptr(fr, C.c_uint8), 56, 64, _lib.ptr(blk, C.c_int32), C.byref(nb)))
    assert blk.min() == 0 and blk.max() + 1 == nb.value
    sizes = np.bincount(blk)
    assert sizes.max() <= 64 and np.all(sizes[np.unique(blk[fr.astype(bool)])] >= 1)
    assert np.all(sizes[blk[~fr.astype(bool)]] == 1)                      # non-free rows: blocks of their own
    free_sizes = np.bincount(blk[fr.astype(bool)])
    assert free_sizes[free_sizes > 0].mean() > 40
    S = sp.csr_matrix((np.ones(p.col.size), p.col, p.rowptr), shape=(p.n, p.n)).tocoo()
    off = S.row != S.col
    frac_compact = np.mean(blk[S.row[off]] == blk[S.col[off]])
    frac_lines = np.mean((S.row[off] // 60) == (S.col[off] // 60))
    assert frac_compact > 1.5 * frac_lines and frac_compact > 0.45       # couplings inside a block: compact vs consecutive rows
    color = np.zeros(p.n, dtype=np.int32)
    nc = C.c_int32()
    _lib.hcheck(lib.amgh_coloring_blockids(C.byref(d), _lib.ptr(fr, C.c_uint8), _lib.ptr(blk, C.c_int32), _lib.ptr(color, C.c_int32), C.byref(nc)))
    assert np.all(color[~fr.astype(bool)] == -1) and np.all(color[fr.astype(bool)] >= 0)
    same = off & (blk[S.row] == blk[S.col]) & (color[S.row] >= 0) & (color[S.col] >= 0)
    assert not np.any(color[S.row[same]] == color[S.col[same]])
    dinv = np.zeros(p.n * 36)
    _lib.hcheck(lib.amgh_hybrid_dinv_block_ids(C.byref(d), _lib.ptr(fr, C.c_uint8), _lib.ptr(blk, C.c_int32), 0, _lib.ptr(dinv, C.c_double)))
    val = p.val.reshape(-1, 6, 6)
    diag = np.zeros((p.n, 6))
    D = np.zeros((p.n, 6, 6))
    for i in range(p.n):
        for k in range(p.rowptr[i], p.rowptr[i + 1]):
            if p.col[k] == i:
                D[i], diag[i] = val[k], np.diag(val[k])
    rng = np.random.default_rng(0)
    for i in rng.choice(np.nonzero(fr)[0], 25, replace=False):
        fac = 1.0
        for l in range(6):
            ad = 0.0
            for k in range(p.rowptr[i], p.rowptr[i + 1]):
                j = p.col[k]
                if blk[j] == blk[i]:
                    continue
                ad += np.sum(np.abs(val[k][l]) / np.sqrt(diag[i, l] * diag[j]))
            fac = max(fac, 0.51 * (1.0 + ad))
        assert np.allclose(dinv[i * 36:(i + 1) * 36].reshape(6, 6), np.linalg.inv(D[i]) / fac, rtol=1e-10, atol=1e-14)


def test_prolongation_types_of_the_reference():
    """ngs_amg_prol_type (vertex_factory_impl.hpp:63-69): piecewise = one unit entry per free row; aux_smoothed = at most
    sp_max_per_row entries, non-negative, rows sum to 1; semi_aux_smoothed (default) = rows whose algebraic neighbours cover at most
    sp_max_per_row_classic aggregates are the rows of (I - omega D^-1 A) P_pw (checked entry by entry), the others are aux rows"""
    from ngsamg_amd.hierarchy import Hierarchy
    p = fem.poisson_fast((23, 19, 17), dirichlet="right|top", jitter=0.2, seed=1)
    A = to_matrix(p)
    free = p.free.astype(bool)
    Hp = Hierarchy(A, p.free, p.coords, dim=3, max_coarse_size=20, prol_type="piecewise")
    Pp = Hp.levels[0].P.to_scipy()
    assert np.all(np.diff(Pp.indptr)[free] == 1) and np.all(Pp.data == 1.0)
    Ha = Hierarchy(A, p.free, p.coords, dim=3, max_coarse_size=20, prol_type="aux_smoothed")
    Pa = Ha.levels[0].P.to_scipy()
    assert np.diff(Pa.indptr).max() <= 3 and Pa.data.min() >= 0.0
    assert np.allclose(np.asarray(Pa.sum(axis=1)).ravel()[free], 1.0, atol=1e-14)
    Hs = Hierarchy(A, p.free, p.coords, dim=3, max_coarse_size=20)                     # default: semi_aux_smoothed
    Hs2 = Hierarchy(A, p.free, p.coords, dim=3, max_coarse_size=20, prol_type="semi_aux_smoothed", sp_max_per_row_classic=5)
    Ps = Hs.levels[0].P.to_scipy()
    assert abs(Ps - Hs2.levels[0].P.to_scipy()).max() == 0.0
    agg = np.asarray(Hs.levels[0].agg)
    assert np.array_equal(agg, np.asarray(Ha.levels[0].agg))                           # same agglomerates, other weights
    As = A.to_scipy().tocsr()
    nc = Ps.shape[1]
    Ppw = sp.csr_matrix((np.ones(free.sum()), (np.nonzero(free)[0], agg[free])), shape=(p.n, nc))
    classic = sp.csr_matrix(Ppw - sp.diags(1.0 / As.diagonal()) @ (As @ Ppw))         # omega = 1 (h1_impl.hpp:324)
    n_classic = 0
    for i in np.nonzero(free)[0]:
        nb = As.indices[As.indptr[i]:As.indptr[i + 1]]
        if np.all(agg[nb] >= 0) and np.unique(agg[nb]).size <= 5 and np.any(agg[nb[nb != i]] == agg[i]):
            n_classic += 1
            assert abs(Ps[i] - classic[i]).max() < 1e-13
        else:
            assert Ps[i].nnz <= 3 and Ps[i].data.min() >= 0.0 and abs(Ps[i].sum() - 1.0) < 1e-14
    assert n_classic > 100                      # (3D: most vertices see more than 5 aggregates and take the aux branch)
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, p.coords, dim=3, prol_type="smoothest")


@pytest.mark.parametrize("rot", [False, True])
def test_block_coloured_gauss_seidel_order_is_inside_the_iteration_tolerance(rot):
    """host data of the block-coloured Gauss-Seidel form (ngsamg_amd.device.block_colored_gs_data: in-block colours, a colouring
    of the block graph, plain diagonal) on an elasticity level with grid-LINE sweep blocks: the oracle's Gauss-Seidel in the order
    (block colour, block, in-block colour) needs at most 15 % more PCG iterations than the reference's sequential order,
    where the hybrid form with the same blocks (couplings between blocks frozen) needs more"""
    from copy import copy
    from ngsamg_amd.device import block_colored_gs_data, hybrid_gs_data
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    nv = 18
    p = fem.elasticity_fast((nv, nv, nv), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    from ngsamg_amd.hierarchy import Hierarchy
    H = Hierarchy(A, p.free, p.coords, dim=3, energy=1, max_coarse_size=20, regularize_cmats=0 if rot else 1)
    lv0 = H.levels[0]
    B = nv                                            # one sweep block = one grid line
    col, nc, bcol, nbc, dinv = block_colored_gs_data(lv0.A, lv0.free, B, lv0.dinv)
    blk = np.arange(lv0.n) // B
    assert nbc >= 2 and np.all(bcol[blk * B] == bcol)                     # constant inside a block
    S = sp.csr_matrix((np.ones(len(lv0.A.col)), np.asarray(lv0.A.col), np.asarray(lv0.A.rowptr)), shape=(lv0.n, lv0.n)).tocoo()
    cross = blk[S.row] != blk[S.col]
    assert np.all(bcol[S.row[cross]] != bcol[S.col[cross]])               # coupled blocks differ: the in-place launches cannot race
    info = [dict(B=B, color=col, n_colors=nc, dinv=dinv, block_of_row=None, block_color=bcol, n_block_colors=nbc)] + [None] * (H.n_levels - 1)
    lv, types = hgs_levels(H.levels, info)
    hcol, hnc, hdinv = hybrid_gs_data(lv0.A, lv0.free, B, pinv=not rot)
    hinfo = [dict(B=B, color=hcol, n_colors=hnc, dinv=hdinv, block_of_row=None)] + [None] * (H.n_levels - 1)
    hlv, htypes = hgs_levels(H.levels, hinfo)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    it_seq = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=200)[1]
    it_bc = Oracle(lv, sm_type=types).pcg(b, tol=1e-8, maxit=200)[1]
    it_hy = Oracle(hlv, sm_type=htypes).pcg(b, tol=1e-8, maxit=200)[1]
    assert it_bc <= int(np.ceil(1.15 * it_seq)), (it_bc, it_seq)
    assert it_bc <= it_hy, (it_bc, it_hy)


def test_galerkin_hook_protocol():
    """amgh_set_galerkin_hook: the hook sees the scalar products of levels with at least min_rows rows; 2 = "not for me" leaves the
    host product in place (same hierarchy); an error code fails the setup; run / fetch come as a pair."""
    import ctypes as C
    from ngsamg_amd import _lib
    from ngsamg_amd.hierarchy import Hierarchy
    lib = _lib.host()
    prob = fem.poisson_fast((12, 11, 10), dirichlet="right")
    A = Matrix(prob.n, prob.n, 1, 1, prob.rowptr, prob.col, prob.val)
    RUN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64))
    FETCH = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double))
    seen = []
    code = [2]

    def run(pt, a, p, res, nr, nnz):
        seen.append(C.cast(a, C.POINTER(_lib.amgh_matrix)).contents.n_rows)
        return code[0]

    run_c, fetch_c = RUN(run), FETCH(lambda *a: 1)
    saved = _lib._device_setup
    try:
        _lib.device_setup(False)
        H0 = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=20, spw=1)
        _lib._device_setup = False          # Hierarchy() must not replace the test's hook
        _lib.hcheck(lib.amgh_set_galerkin_hook(C.cast(run_c, C.c_void_p), C.cast(fetch_c, C.c_void_p), 100))
        H1 = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=20, spw=1)
        sizes = [l.A.n_rows for l in H0.levels[:-1]]
        assert seen == [n for n in sizes if n >= 100] and len(seen) >= 1
        for a, b in zip(H0.levels, H1.levels):
            assert np.array_equal(a.A.col, b.A.col) and np.array_equal(a.A.val, b.A.val)
        code[0] = 1
        with pytest.raises(NgsAMGError, match="galerkin hook failed"):
            Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=20, spw=1)
        assert lib.amgh_set_galerkin_hook(C.cast(run_c, C.c_void_p), None, 0) != 0
    finally:
        lib.amgh_set_galerkin_hook(None, None, 0)
        _lib._device_setup = saved


def test_device_setup_without_a_gpu_keeps_the_host_product():
    """Hierarchy() asks for the device Galerkin product on every setup; without a GPU that is a no-op, insisting on it raises"""
    import torch
    from ngsamg_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the hook is installed (tests/test_gpu_spgemm.py)")
    saved = _lib._device_setup
    try:
        _lib._device_setup = None
        assert _lib.device_setup() is False
        with pytest.raises(NgsAMGError, match="no HIP device"):
            _lib.device_setup(True)
    finally:
        _lib._device_setup = saved


@pytest.mark.parametrize("rot", [False, True])
def test_edge_matrix_prolongation_of_the_reference(rot):
    """ngs_amg_edge_mats: the setup carries the elasticity energy's edge matrices (finest level as BuildAlgMesh_ALG_blk,
    elasticity_pc_impl.hpp:409-505; coarse levels as AttachedEED::map_data, elasticity_impl.hpp:23-78) and builds the
    matrix-valued smoothed prolongation (SemiAuxSProlMap, vertex_factory_impl.hpp:1836-2290).  Properties the reference's
    construction guarantees: general BS x BS blocks (not w Q(t)); every row of a free vertex reproduces the rigid-body modes on
    every level (aux rows by construction, classic rows away from the clamped face because A annihilates them there); Galerkin
    coarse operators; and the budgets of tests/elasticity/mdim/simple/test_3d_lo.py (40 iterations, tol 1e-6) at its mesh size."""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    shape, ext = ((9, 5, 5), (2.0, 1.0, 1.0)) if rot else ((41, 5, 5), (10.0, 1.0, 1.0))
    p = fem.elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.0, rotations=rot, extent=ext)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    kw = dict(dim=3, energy=1, max_coarse_size=10, regularize_cmats=0 if rot else 1)
    H = Hierarchy(A, p.free, p.coords, edge_mats=1, **kw)
    H0 = Hierarchy(A, p.free, p.coords, **kw)
    assert H.n_levels >= 3 and all(L.bs == 6 for L in H.levels[1:])
    rng = np.random.default_rng(2)
    t, w = rng.standard_normal(3), rng.standard_normal(3)
    general_blocks = False
    for l in range(H.n_levels - 1):
        Lf, Lc = H.levels[l], H.levels[l + 1]
        P = Lf.P.to_scipy()
        coarse = np.concatenate([t + np.cross(w, Lc.coords), np.tile(w, (Lc.n, 1))], axis=1).ravel()
        fu = t + np.cross(w, Lf.coords)
        fine = fu if Lf.bs == 3 else np.concatenate([fu, np.tile(w, (Lf.n, 1))], axis=1)
        got = (P @ coarse).reshape(Lf.n, -1)
        free = Lf.free.astype(bool)
        err = np.abs(got[free] - fine[free]).max(axis=1)
        scale = max(1.0, np.abs(fine).max())
        # a classic row is pw_i - omega D^+ (A u)_i for the rigid-body mode u: exact where the level matrix annihilates u, i.e. away
        # from the clamped face and its images on the coarse levels (the reference gives up the kernel there,
        # vertex_factory_impl.hpp:2108-2117); level 0 of the displacement-only problem is aux only: exact everywhere
        if l == 0 and not rot:
            assert err.max() < 1e-10 * scale
        else:
            Af = Lf.A.to_scipy()
            res = np.abs((Af @ (fine * free[:, None]).ravel()).reshape(Lf.n, -1)).max(axis=1)
            sees_clamp = res > 1e-9 * abs(Af).max() * scale
            assert 0 < (~sees_clamp[free]).sum()
            assert err[~sees_clamp[free]].max() < 1e-8 * scale
        Ac = Lc.A.to_scipy()
        assert abs(Ac - Ac.T).max() < 1e-10 * abs(Ac).max()
        assert abs(Ac - P.T @ Lf.A.to_scipy() @ P).max() < 1e-10 * abs(Ac).max()
        # blocks beyond w Q(t): the displacement-displacement part of some block is not a multiple of the identity
        blk = np.asarray(Lf.P.val).reshape(-1, Lf.P.br, Lf.P.bc)[:, :3, :3]
        off = np.abs(blk - np.eye(3) * blk[:, :1, :1]).max()
        general_blocks = general_blocks or off > 1e-6
    assert general_blocks
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    it = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-6, maxit=200)[1]
    it0 = Oracle(H0.levels, sm_type="gs").pcg(b, tol=1e-6, maxit=200)[1]
    assert it <= 40, it                       # the reference's budget
    assert it <= it0 + 5, (it, it0)           # the scalar-weight rule w Q(t) of the default setup on the same problem


def test_edge_matrix_option_errors_and_2d():
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    p = fem.poisson_fast((9, 9), dirichlet="left")
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, None, dim=2, energy=0, edge_mats=1)             # an option of the elasticity energy
    q = fem.elasticity_fast((41, 9), dirichlet="left", mu=1.0, lam=0.0, rotations=False, extent=(5.0, 1.0))
    B = Matrix(q.n, q.n, q.bs, q.bs, q.rowptr, q.col, q.val)
    with pytest.raises(NgsAMGError):
        Hierarchy(B, q.free, q.coords, dim=2, energy=1, edge_mats=1, spw=0)
    H = Hierarchy(B, q.free, q.coords, dim=2, energy=1, edge_mats=1, max_coarse_size=20, regularize_cmats=1)
    assert H.levels[0].bs == 2 and all(L.bs == 3 for L in H.levels[1:])
    # 2D rigid-body modes: u = t + w (-y, x), rotation w
    rng = np.random.default_rng(5)
    t, w = rng.standard_normal(2), float(rng.standard_normal())
    Lf, Lc = H.levels[0], H.levels[1]
    rb = lambda X: np.concatenate([t + w * np.stack([-X[:, 1], X[:, 0]], axis=1), np.full((len(X), 1), w)], axis=1)
    got = (Lf.P.to_scipy() @ rb(Lc.coords).ravel()).reshape(Lf.n, 2)
    free = Lf.free.astype(bool)
    assert np.abs(got[free] - rb(Lf.coords)[free, :2]).max() < 1e-10
    b = rng.standard_normal(q.n * q.bs) * np.repeat(q.free, q.bs)
    assert Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-6, maxit=200)[1] <= 50   # budget of tests/elasticity/mdim/simple/test_2d_lo.py


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("robust", [0, 1])
def test_edge_matrix_setup_on_material_jumps_and_robust_soc(rot, robust):
    """edge_mats (+ crs_robust: the energy-based strength of connection of the SPW rounds, CalcRobSOC with neighbour boost,
    agglomerator_utils.hpp:598-927, picked as FindNeib3Step does with robustPick, spw_agg_impl.hpp:637-775) on the stiff-inclusion
    problem of the reference's tests/elasticity/mdim/jump/test_2d_jump_lo.py (mu jump 1e4, max_coarse_size 10, budget 50) and on
    its 3D beams (tests/elasticity/mdim/simple/test_3d_lo.py, budget 40): budgets met, rigid-body modes reproduced, and the
    robust rule really changes the aggregates"""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle

    def coef(X):
        x, y = X[..., 0], X[..., 1]
        inner = ((np.abs(x - 0.3) < 0.1) | (np.abs(x - 0.7) < 0.1)) & ((np.abs(y - 0.3) < 0.1) | (np.abs(y - 0.7) < 0.1))
        return np.where(inner, 1e4, 1.0)

    rng = np.random.default_rng(3)
    p = fem.elasticity_fast((41, 41), dirichlet="left", mu=1.0, lam=0.0, rotations=rot, coef=coef)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    kw = dict(dim=2, energy=1, max_coarse_size=10, regularize_cmats=0 if rot else 1, edge_mats=1)
    H = Hierarchy(A, p.free, p.coords, crs_robust=robust, **kw)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    assert Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-6, maxit=200)[1] <= 50
    if robust:
        Hs = Hierarchy(A, p.free, p.coords, crs_robust=0, **kw)
        assert not np.array_equal(np.asarray(H.levels[0].agg), np.asarray(Hs.levels[0].agg))
    shape, ext = ((9, 5, 5), (2.0, 1.0, 1.0)) if rot else ((41, 5, 5), (10.0, 1.0, 1.0))
    q = fem.elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.0, rotations=rot, extent=ext)
    B = Matrix(q.n, q.n, q.bs, q.bs, q.rowptr, q.col, q.val)
    H3 = Hierarchy(B, q.free, q.coords, dim=3, energy=1, max_coarse_size=10, regularize_cmats=0 if rot else 1, edge_mats=1, crs_robust=robust)
    b3 = rng.standard_normal(q.n * q.bs) * np.repeat(q.free, q.bs)
    assert Oracle(H3.levels, sm_type="gs").pcg(b3, tol=1e-6, maxit=200)[1] <= 40
    Lf, Lc = H3.levels[0], H3.levels[1]
    t, w = rng.standard_normal(3), rng.standard_normal(3)
    coarse = np.concatenate([t + np.cross(w, Lc.coords), np.tile(w, (Lc.n, 1))], axis=1).ravel()
    fu = t + np.cross(w, Lf.coords)
    fine = fu if Lf.bs == 3 else np.concatenate([fu, np.tile(w, (Lf.n, 1))], axis=1)
    got = (Lf.P.to_scipy() @ coarse).reshape(Lf.n, -1)
    free = Lf.free.astype(bool)
    err = np.abs(got[free] - fine[free]).max(axis=1)
    if not rot:
        assert err.max() < 1e-9
    else:
        assert np.median(err) < 1e-9
    with pytest.raises(NgsAMGError):
        Hierarchy(B, q.free, q.coords, dim=3, energy=1, crs_robust=1)          # needs the edge matrices


def test_aggregate_wide_stability_check_of_the_spw_rounds():
    """ngs_amg_spw_cbs (checkBigSOC, spw_agg.hpp:31, off by default in the reference): from the second pairing round on a pair must
    pass AggregateWideStabilityCheck (agglomerator_utils.hpp:392-539) on the union of its base-level members.  With full-rank edge
    matrices (rotations: E = x I) the pairs the robust rule picks pass it; with the rank-one springs of a displacement-only level
    the replacement matrix of a few vertices is far weaker than their aux diagonals and the check keeps aggregates small -- slower
    coarsening, still a convergent hierarchy inside the reference's budget."""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    rng = np.random.default_rng(4)
    for rot, shape, ext in ((True, (9, 5, 5), (2.0, 1.0, 1.0)), (False, (21, 5, 5), (5.0, 1.0, 1.0))):
        p = fem.elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.0, rotations=rot, extent=ext)
        A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
        kw = dict(dim=3, energy=1, max_coarse_size=10, regularize_cmats=0 if rot else 1, edge_mats=1, crs_robust=1)
        H0 = Hierarchy(A, p.free, p.coords, **kw)
        H1 = Hierarchy(A, p.free, p.coords, spw_cbs=1, **kw)
        assert H1.levels[1].n >= H0.levels[1].n
        if rot:
            assert H1.levels[1].n <= 1.2 * H0.levels[1].n
        else:
            assert H1.levels[1].n > 1.5 * H0.levels[1].n
        b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
        assert Oracle(H1.levels, sm_type="gs").pcg(b, tol=1e-6, maxit=200)[1] <= 40
        for l in range(H1.n_levels - 1):
            Lf, Lc = H1.levels[l], H1.levels[l + 1]
            P, Ac = Lf.P.to_scipy(), Lc.A.to_scipy()
            assert abs(Ac - P.T @ Lf.A.to_scipy() @ P).max() < 1e-10 * abs(Ac).max()
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, p.coords, dim=3, energy=1, edge_mats=1, spw_cbs=1)          # belongs to crs_robust


def test_prolongation_improve_steps_keep_the_graph_and_the_kernel():
    """ngs_amg_sp_improve_its (vertex_factory_impl.hpp:1745-1831, 2350-2420; off by default): P_i -= omega D^+ (A P)_i with the
    entries outside the row's pattern moved to the row's own aggregate.  The graph of P does not change; a row whose matrix row
    annihilates the kernel (constants / rigid-body modes: away from Dirichlet vertices) still reproduces it exactly; the Galerkin
    relation holds for the improved P; on the elasticity problems the Gauss-Seidel PCG count does not get worse."""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    rng = np.random.default_rng(6)
    p = fem.poisson_fast((16, 16, 16), dirichlet="right|top", jitter=0.2, seed=1)
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    H0 = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=20)
    H2 = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=20, sp_improve_its=2)
    P0, P2 = H0.levels[0].P, H2.levels[0].P
    assert np.array_equal(np.asarray(P0.rowptr), np.asarray(P2.rowptr)) and np.array_equal(np.asarray(P0.col), np.asarray(P2.col))
    assert not np.allclose(np.asarray(P0.val), np.asarray(P2.val))
    free = p.free.astype(bool)
    Af = H2.levels[0].A.to_scipy()
    sees = np.abs(Af @ free.astype(float)) > 1e-10 * abs(Af).max()
    sees = sees | ((abs(Af) @ sees.astype(float)) > 0)          # the second step reads the first step's rows of the neighbours
    rs = np.asarray(P2.to_scipy().sum(axis=1)).ravel()
    assert (free & ~sees).sum() > 100 and np.abs(rs[free & ~sees] - 1.0).max() < 1e-12
    A1 = H2.levels[1].A.to_scipy()
    Ps = P2.to_scipy()
    assert abs(A1 - Ps.T @ Af @ Ps).max() < 1e-10 * abs(A1).max()
    b = rng.standard_normal(p.n) * p.free
    assert Oracle(H2.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=200)[1] <= Oracle(H0.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=200)[1] + 2
    for rot, em in ((False, 0), (False, 1), (True, 1)):
        q = fem.elasticity_fast((12, 12, 12), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
        B = Matrix(q.n, q.n, q.bs, q.bs, q.rowptr, q.col, q.val)
        kw = dict(dim=3, energy=1, max_coarse_size=20, regularize_cmats=0 if rot else 1, edge_mats=em)
        G0 = Hierarchy(B, q.free, q.coords, **kw)
        G2 = Hierarchy(B, q.free, q.coords, sp_improve_its=2, **kw)
        L0, L1 = G2.levels[0], G2.levels[1]
        assert np.array_equal(np.asarray(G0.levels[0].P.col), np.asarray(L0.P.col))
        t, w = rng.standard_normal(3), rng.standard_normal(3)
        coarse = np.concatenate([t + np.cross(w, L1.coords), np.tile(w, (L1.n, 1))], axis=1).ravel()
        fu = t + np.cross(w, L0.coords)
        fine = fu if L0.bs == 3 else np.concatenate([fu, np.tile(w, (L0.n, 1))], axis=1)
        fr = L0.free.astype(bool)
        Aq = L0.A.to_scipy()
        res = np.abs((Aq @ (fine * fr[:, None]).ravel()).reshape(L0.n, -1)).max(axis=1)
        bad = ~fr | (res >= 1e-9 * abs(Aq).max() * max(1.0, np.abs(fine).max()))
        Sq = sp.csr_matrix((np.ones(len(L0.A.col)), np.asarray(L0.A.col), np.asarray(L0.A.rowptr)), shape=(L0.n, L0.n))
        ok = ~(bad | ((Sq @ bad.astype(float)) > 0))
        got = (L0.P.to_scipy() @ coarse).reshape(L0.n, -1)
        assert ok.sum() > 100 and np.abs(got[ok] - fine[ok]).max() < 1e-9 * max(1.0, np.abs(fine).max())
        bb = rng.standard_normal(q.n * q.bs) * np.repeat(q.free, q.bs)
        it0 = Oracle(G0.levels, sm_type="gs").pcg(bb, tol=1e-8, maxit=200)[1]
        it2 = Oracle(G2.levels, sm_type="gs").pcg(bb, tol=1e-8, maxit=200)[1]
        assert it2 <= it0, (rot, em, it2, it0)
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, p.coords, dim=3, energy=0, sp_improve_its=1, spw=0, enable_multistep=1)


@pytest.mark.parametrize("energy", [0, 1])
def test_prol_only_step_equals_the_first_step_of_the_full_setup(energy):
    """amgh_options.prol_only (what the rank-partitioned setup calls per level): one coarsening step that returns P, the aggregates
    and the coarse coordinates and skips P^T, the Galerkin product, smoother data and the dense coarse inverse -- bit-identical to
    level 0 of the full setup"""
    from ngsamg_amd.hierarchy import Hierarchy
    if energy == 0:
        p = fem.poisson_fast((14, 13, 12), dirichlet="right|top", jitter=0.2, seed=1)
        kw = dict(dim=3, energy=0)
    else:
        p = fem.elasticity_fast((9, 8, 7), dirichlet="left", mu=1.0, lam=0.5, rotations=False)
        kw = dict(dim=3, energy=1, regularize_cmats=1)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, max_coarse_size=10, **kw)
    Hp = Hierarchy(A, p.free, p.coords, max_coarse_size=10, prol_only=1, **kw)
    assert Hp.n_levels == 2 and Hp.levels[0].PT is None and Hp.coarse_n == 0
    for name in ("rowptr", "col", "val"):
        assert np.array_equal(np.asarray(getattr(H.levels[0].P, name)), np.asarray(getattr(Hp.levels[0].P, name)))
    assert np.array_equal(np.asarray(H.levels[0].agg), np.asarray(Hp.levels[0].agg))
    assert np.array_equal(H.levels[1].coords, Hp.levels[1].coords)
    L1 = Hp.levels[1]
    assert L1.n == H.levels[1].n and L1.bs == H.levels[1].bs and L1.A.nnz == 0 and L1.dinv.shape[0] == L1.n * L1.bs ** 2
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, p.coords, prol_only=1, spw=0, enable_multistep=1, **kw)


def test_robust_pair_soc_is_the_smallest_generalised_eigenvalue():
    """amgh_robust_pair_soc (CalcRobustPairSOC, agglomerator_utils.hpp:763-841) against scipy: full-rank C, rank-deficient C (the
    kernel of C is projected out, where E may be anything), scaling behaviour, the 1 x 1 case"""
    import ctypes as C
    import scipy.linalg as sla
    from ngsamg_amd import _lib
    lib = _lib.host()

    def soc(Cm, Em):
        n = Cm.shape[0]
        out = C.c_double()
        Cm, Em = np.ascontiguousarray(Cm, dtype=np.float64), np.ascontiguousarray(Em, dtype=np.float64)
        _lib.hcheck(lib.amgh_robust_pair_soc(n, _lib.ptr(Cm, C.c_double), _lib.ptr(Em, C.c_double), C.byref(out)))
        return out.value

    rng = np.random.default_rng(9)
    for n in (2, 3, 6):
        X, Y = rng.standard_normal((n, n)), rng.standard_normal((n, n))
        Cm, Em = X @ X.T + 0.1 * np.eye(n), Y @ Y.T
        ref = max(0.0, sla.eigh(Em, Cm, eigvals_only=True)[0])
        assert abs(soc(Cm, Em) - ref) <= 1e-9 * max(1.0, ref)
        assert abs(soc(3.0 * Cm, Em) - ref / 3.0) <= 1e-9 and abs(soc(Cm, 5.0 * Em) - 5.0 * ref) <= 1e-8 * max(1.0, ref)
        # rank-deficient C: only range(C) counts
        k = n - 1
        U = np.linalg.qr(rng.standard_normal((n, n)))[0]
        lam = np.concatenate([rng.uniform(0.5, 2.0, k), [0.0]])
        Cd = (U * lam) @ U.T
        W = U[:, :k] / np.sqrt(lam[:k])
        ref = max(0.0, np.linalg.eigvalsh(W.T @ Em @ W)[0])
        assert abs(soc(Cd, Em) - ref) <= 1e-8 * max(1.0, ref)
        assert soc(np.zeros((n, n)), Em) == 0.0
    assert abs(soc(np.array([[4.0]]), np.array([[2.0]])) - 0.5) < 1e-14
    with pytest.raises(NgsAMGError):
        soc(np.eye(7), np.eye(7))


def test_robust_agglomeration_switches_of_the_reference():
    """ngs_amg_spw_pick_robust = False (the scalar order decides, the robust number only vetoes) and ngs_amg_spw_neib_boost = False
    (spw_agg.hpp:26-27, 55-56): both give valid, different hierarchies inside the reference's budget on its 3D beam"""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    p = fem.elasticity_fast((21, 5, 5), dirichlet="left", mu=1.0, lam=0.0, rotations=True, extent=(5.0, 1.0, 1.0))
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    kw = dict(dim=3, energy=1, max_coarse_size=10, regularize_cmats=0, edge_mats=1, crs_robust=1)
    b = np.random.default_rng(8).standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    aggs = {}
    for name, extra in (("default", {}), ("veto", {"spw_pick_robust": 0}), ("noboost", {"spw_neib_boost": 0})):
        H = Hierarchy(A, p.free, p.coords, **kw, **extra)
        aggs[name] = np.asarray(H.levels[0].agg).copy()
        assert H.n_levels >= 3
        assert Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-6, maxit=200)[1] <= 40
        Lf, Lc = H.levels[0], H.levels[1]
        P = Lf.P.to_scipy()
        assert abs(Lc.A.to_scipy() - P.T @ Lf.A.to_scipy() @ P).max() < 1e-10 * abs(Lc.A.to_scipy()).max()
    assert not np.array_equal(aggs["default"], aggs["veto"])
    assert not np.array_equal(aggs["default"], aggs["noboost"])


def test_spw_pick_avg_and_diag_stab_boost():
    """ngs_amg_spw_pick_avg (min | geom | harm | alg | max; default geom) changes the scalar filter of the pairing, so the aggregates of
    a problem with a coefficient jump (where the two vertices' scales differ) move; on a quasi-uniform problem with equal scales
    every average gives the same aggregates.  ngs_amg_spw_diag_stab_boost acts on the carried aux diagonals of crs_robust."""
    from ngsamg_amd.hierarchy import Hierarchy

    def coef(X):
        return np.where(X[..., 0] > 0.5, 1e3, 1.0)

    p = fem.poisson_fast((21, 21), dirichlet="left", coef=coef) if "coef" in fem.poisson_fast.__code__.co_varnames else None
    if p is not None:
        A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
        aggs = {a: np.asarray(Hierarchy(A, p.free, p.coords, dim=2, energy=0, max_coarse_size=10, spw_pick_avg=a).levels[0].agg).copy()
                for a in ("min", "geom", "max")}
        assert not (np.array_equal(aggs["min"], aggs["geom"]) and np.array_equal(aggs["max"], aggs["geom"]))
    q = fem.elasticity_fast((13, 5, 5), dirichlet="left", mu=1.0, lam=0.0, rotations=True, extent=(3.0, 1.0, 1.0))
    B = Matrix(q.n, q.n, q.bs, q.bs, q.rowptr, q.col, q.val)
    kw = dict(dim=3, energy=1, max_coarse_size=10, regularize_cmats=0, edge_mats=1, crs_robust=1)
    a0 = np.asarray(Hierarchy(B, q.free, q.coords, spw_diag_stab_boost=0.0, **kw).levels[0].agg).copy()
    a1 = np.asarray(Hierarchy(B, q.free, q.coords, spw_diag_stab_boost=1.0, **kw).levels[0].agg).copy()
    assert a0.max() > 0 and a1.max() > 0 and not np.array_equal(a0, a1)
    with pytest.raises(NgsAMGError):
        Hierarchy(B, q.free, q.coords, spw_pick_avg="median", **kw)
    with pytest.raises(NgsAMGError):
        Hierarchy(B, q.free, q.coords, spw_pick_avg=7, **kw)


@pytest.mark.parametrize("name", ["elast3d_5_bs3_edge_mats", "elast3d_4_bs6_edge_mats"])
def test_edge_matrix_setup_reproduces_its_golden_fixture(name):
    """the committed fixtures of the edge-matrix setup (tests/golden/make_golden.py: edge_mats; edge_mats + crs_robust +
    sp_improve_its) pin its prolongations and coarse operators against regressions"""
    import importlib.util
    import os
    from tests import golden_io
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(golden_io.GOLDEN_DIR, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    p, H = mg.build(mg.CASES[name])
    z, levels = golden_io.load(name)
    assert H.n_levels == len(levels)
    for L, G in zip(H.levels, levels):
        for tag in ("A", "P"):
            M, F = getattr(L, tag), getattr(G, tag)
            assert (M is None) == (F is None)
            if M is None:
                continue
            assert np.array_equal(np.asarray(M.rowptr), np.asarray(F.rowptr)) and np.array_equal(np.asarray(M.col), np.asarray(F.col))
            assert np.allclose(np.asarray(M.val), np.asarray(F.val), rtol=1e-10, atol=1e-12 * np.abs(np.asarray(F.val)).max())


def test_coarsest_level_regularisation_of_the_reference():
    """regularize_cmats on an elasticity hierarchy: before the coarsest matrix is inverted its diagonal blocks are regularised as the
    reference's CoarseLevelInv does (RegularizeMatrix / RegTM<0,6,6>, elasticity_pc_impl.hpp:710-764, utils_denseLA.hpp:1198-1233):
    the smallest non-zero eigenvalue of a block is added along the block's kernel.  Two-vertex aggregates of a displacement-only
    beam (no rotation about the pair's axis) give singular 6x6 blocks; the returned inverse is the inverse of the regularised
    matrix, and a hierarchy whose coarsest blocks are regular is inverted exactly as without the rule."""
    from ngsamg_amd.hierarchy import Hierarchy
    p = fem.elasticity_fast((7, 2, 2), dirichlet="left", mu=1.0, lam=0.0, rotations=False)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, dim=3, energy=1, max_levels=2, max_coarse_size=1, enable_sp=0, regularize_cmats=1, spw_rounds=1)
    assert "diagonal block(s) regularised" in H.log
    L = H.levels[-1]
    Ac = L.A.to_scipy().toarray()
    n, bs = L.n, L.bs
    assert bs == 6 and H.coarse_n == n * bs
    Areg = Ac.copy()
    nsing = 0
    for i in range(n):
        blk = Ac[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs]
        ev, V = np.linalg.eigh(0.5 * (blk + blk.T))
        eps = max(1e-15, 1e-12 * ev[-1])
        zero = ev <= eps
        if zero.any():
            nsing += 1
            mn = ev[~zero].min()
            Areg[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs] += mn * (V[:, zero] @ V[:, zero].T)
    assert nsing > 0
    inv = np.asarray(H.coarse_inv).reshape(n * bs, n * bs)
    assert np.abs(inv @ Areg - np.eye(n * bs)).max() < 1e-8
    assert np.linalg.eigvalsh(0.5 * (Areg + Areg.T))[0] > 0          # the regularised matrix is positive definite: Cholesky, no fallback
    assert "pseudo-inverse used" not in H.log
    # regular coarsest blocks: nothing changes
    q = fem.elasticity_fast((9, 5, 5), dirichlet="left", mu=1.0, lam=0.0, rotations=False, extent=(2.0, 1.0, 1.0))
    B = Matrix(q.n, q.n, q.bs, q.bs, q.rowptr, q.col, q.val)
    G = Hierarchy(B, q.free, q.coords, dim=3, energy=1, max_coarse_size=10, regularize_cmats=1)
    assert "regularised" not in G.log
    Lc = G.levels[-1]
    Acc = Lc.A.to_scipy().toarray()
    assert np.abs(np.asarray(G.coarse_inv).reshape(Acc.shape) @ Acc - np.eye(Acc.shape[0])).max() < 1e-8


def test_carried_alg_mesh_gives_an_equivalent_hierarchy():
    """carry_mesh: coarse alg-meshes by contraction of the finer mesh (the reference's way) instead of the Galerkin matrix's graph:
    level 0 is untouched, the coarse levels coarsen at the same rate, the hierarchy is Galerkin and as good a preconditioner"""
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    p = fem.poisson_fast((20, 20, 20), dirichlet="right|top", jitter=0.2, seed=1)
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    H0 = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=20)
    H1 = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=20, carry_mesh=1)
    assert np.array_equal(np.asarray(H0.levels[0].P.val), np.asarray(H1.levels[0].P.val)) and H0.levels[1].n == H1.levels[1].n
    assert H1.n_levels >= 3 and not np.array_equal(np.asarray(H0.levels[1].agg), np.asarray(H1.levels[1].agg))
    assert abs(H1.levels[2].n - H0.levels[2].n) <= 0.1 * H0.levels[2].n
    for l in range(H1.n_levels - 1):
        P = H1.levels[l].P.to_scipy()
        Ac = H1.levels[l + 1].A.to_scipy()
        assert abs(Ac - P.T @ H1.levels[l].A.to_scipy() @ P).max() < 1e-10 * abs(Ac).max()
    b = np.random.default_rng(1).standard_normal(p.n) * p.free
    it0 = Oracle(H0.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=100)[1]
    it1 = Oracle(H1.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=100)[1]
    assert abs(it1 - it0) <= 2
    with pytest.raises(NgsAMGError):
        Hierarchy(A, p.free, p.coords, dim=3, energy=0, carry_mesh=1, spw=0)


@pytest.mark.parametrize("rot", [False, True])
def test_default_gpu_gauss_seidel_order_of_block_levels_is_inside_the_tolerance(rot):
    """the order data DeviceAMGMatrix hands to the device for sm_type = hgs (ngsamg_amd.device.hierarchy_desc: host code) on the
    elasticity toy of tests/test_gpu_hgs.py, replayed by the oracle: PCG iterations within ceil(1.15 x) of the sequential order"""
    from ngsamg_amd.device import hierarchy_desc
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    from tests.hgs_oracle import hgs_levels
    p, H = elasticity_case((16, 14, 12), rotations=rot, max_coarse_size=10)
    _, _, hgs = hierarchy_desc(H, "hgs")
    lv, types = hgs_levels(H.levels, hgs)
    fr = np.repeat(p.free, p.bs).astype(np.float64)
    for seed in range(3):
        b = np.random.default_rng(seed).standard_normal(p.n * p.bs) * fr
        it_h = Oracle(lv, sm_type=types).pcg(b, tol=1e-8, maxit=200)[1]
        it_seq = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=200)[1]
        assert it_h <= int(np.ceil(1.15 * it_seq)), (it_h, it_seq)
