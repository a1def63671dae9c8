"""Setup products on the device (amgx_spgemm / amgx_galerkin, csrc/device/spgemm.hpp) against the host library's products
(amgh_matmul, restrict_matrix; reference MatMultABImpl utils_sparseMM.cpp:107-238, RestrictMatrix utils_sparseMM.hpp:93-109):
row pointers, columns and VALUES bit for bit -- both accumulate entry (i, j) as c = fma(a_ik, b_kj, c) over k ascending."""
import numpy as np
import pytest
import scipy.sparse as sp

from ngsamg_amd import _lib, fem
from ngsamg_amd._lib import Matrix
from ngsamg_amd.device import device_galerkin, device_spmm
from ngsamg_amd.hierarchy import Hierarchy
from ngsamg_amd.NgsAMG import SparseMM

pytestmark = pytest.mark.gpu


def _rand_csr(rng, n_rows, n_cols, row_len):
    rowptr = np.zeros(n_rows + 1, dtype=np.int64)
    cols = []
    for i in range(n_rows):
        L = max(0, min(int(row_len(i)), n_cols))
        c = np.sort(rng.choice(n_cols, size=L, replace=False)).astype(np.int32)
        cols.append(c)
        rowptr[i + 1] = rowptr[i] + L
    col = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int32)
    val = rng.standard_normal(len(col)) * 10.0 ** rng.integers(-3, 4, size=len(col))
    return Matrix(n_rows, n_cols, 1, 1, rowptr, col.astype(np.int32), val)


def _rand_bcsr(rng, n_rows, n_cols, br, bc, row_len):
    S = _rand_csr(rng, n_rows, n_cols, row_len)
    val = rng.standard_normal(S.nnz * br * bc) * 10.0 ** rng.integers(-3, 4, size=S.nnz * br * bc)
    return Matrix(n_rows, n_cols, br, bc, S.rowptr, S.col, val)


def _same(C, D):
    assert C is not None
    assert (C.n_rows, C.n_cols, C.br, C.bc) == (D.n_rows, D.n_cols, D.br, D.bc)
    assert np.array_equal(np.asarray(C.rowptr), np.asarray(D.rowptr))
    assert np.array_equal(np.asarray(C.col), np.asarray(D.col))
    assert np.array_equal(np.asarray(C.val).view(np.uint64), np.asarray(D.val).view(np.uint64))      # bit for bit


@pytest.mark.parametrize("case", ["short", "mixed", "wave", "workgroup", "empty"])
def test_device_product_equals_host_product_bit_for_bit(case):
    rng = np.random.default_rng({"short": 1, "mixed": 2, "wave": 3, "workgroup": 4, "empty": 5}[case])
    if case == "short":            # every row in the 16-lane class (<= 256 products)
        A = _rand_csr(rng, 3000, 2500, lambda i: rng.integers(0, 12))
        B = _rand_csr(rng, 2500, 4000, lambda i: rng.integers(0, 18))
    elif case == "mixed":          # all three size classes in one product, duplicate-heavy (few columns)
        A = _rand_csr(rng, 700, 900, lambda i: [3, 40, 200][i % 3])
        B = _rand_csr(rng, 900, 600, lambda i: rng.integers(5, 40))
    elif case == "wave":           # one wave per row (256 < products <= 2048)
        A = _rand_csr(rng, 500, 800, lambda i: rng.integers(20, 50))
        B = _rand_csr(rng, 800, 5000, lambda i: rng.integers(15, 40))
    elif case == "workgroup":      # one workgroup per row (2048 < products <= 8192), up to several thousand distinct columns
        A = _rand_csr(rng, 60, 400, lambda i: rng.integers(60, 100))
        B = _rand_csr(rng, 400, 20000, lambda i: rng.integers(40, 80))
    else:                          # empty rows, empty B rows, a matrix without entries
        A = _rand_csr(rng, 300, 200, lambda i: 0 if i % 4 == 0 else rng.integers(0, 5))
        B = _rand_csr(rng, 200, 150, lambda i: 0 if i % 3 == 0 else rng.integers(0, 6))
    _same(device_spmm(A, B), SparseMM(A, B))
    if case == "empty":
        Z = _rand_csr(rng, 50, 200, lambda i: 0)
        _same(device_spmm(Z, B), SparseMM(Z, B))


def test_products_the_device_does_not_take():
    rng = np.random.default_rng(7)
    A = _rand_csr(rng, 4, 300, lambda i: 120)
    B = _rand_csr(rng, 300, 40000, lambda i: 100)          # 12000 products per row
    assert device_spmm(A, B) is None
    Ab = Matrix(2, 2, 7, 7, np.array([0, 1, 2], dtype=np.int64), np.array([0, 1], dtype=np.int32), np.arange(98.0))
    assert device_spmm(Ab, Ab) is None                      # result blocks beyond 6 x 6


@pytest.mark.parametrize("shape", [(6, 6, 6), (6, 3, 6), (3, 3, 3)])
def test_block_rows_beyond_the_lds_tables_accumulate_in_the_result(shape):
    br, bk, bc = shape
    rng = np.random.default_rng(11)
    A = _rand_bcsr(rng, 40, 60, br, bk, lambda i: rng.integers(30, 50))
    B = _rand_bcsr(rng, 60, 3000, bk, bc, lambda i: rng.integers(40, 70))      # ~1500 distinct blocks per row, every one hit several times
    _same(device_spmm(A, B), SparseMM(A, B))


@pytest.mark.parametrize("shape", [(6, 3, 3), (6, 3, 6), (6, 6, 6), (3, 2, 2), (3, 2, 3), (3, 3, 3), (2, 2, 2), (1, 3, 6), (6, 3, 1)])
def test_block_products_equal_the_host_product_bit_for_bit(shape):
    br, bk, bc = shape
    rng = np.random.default_rng(100 * br + 10 * bk + bc)
    # rows with few distinct columns (four rows per workgroup), with 50 ... 90 and with 100 ... 180 (one row per workgroup, the
    # larger tables), empty rows
    A = _rand_bcsr(rng, 400, 300, br, bk, lambda i: [0, 2, 6, 14][i % 4])
    B = _rand_bcsr(rng, 300, 2000, bk, bc, lambda i: rng.integers(3, 14))
    _same(device_spmm(A, B), SparseMM(A, B))
    A = _rand_bcsr(rng, 50, 200, br, bk, lambda i: rng.integers(20, 40))
    B = _rand_bcsr(rng, 200, 150, bk, bc, lambda i: rng.integers(5, 12))          # duplicate-heavy: every column many times
    _same(device_spmm(A, B), SparseMM(A, B))


@pytest.mark.parametrize("rot", [False, True])
def test_elasticity_hierarchy_is_identical_with_the_device_hook(rot):
    prob = fem.elasticity_fast((14, 12, 11), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
    A = Matrix(prob.n, prob.n, prob.bs, prob.bs, prob.rowptr, prob.col, prob.val)
    kw = dict(dim=3, energy=1, max_coarse_size=50, regularize_cmats=0 if rot else 1, spw=1)
    try:
        assert _lib.device_setup(False) is False
        H0 = Hierarchy(A, prob.free, prob.coords, **kw)
        assert _lib.device_setup(True, min_rows=0) is True
        H1 = Hierarchy(A, prob.free, prob.coords, **kw)
    finally:
        _lib._device_setup = None
        _lib.device_setup()
    assert len(H0.levels) == len(H1.levels) >= 3
    for a, b in zip(H0.levels, H1.levels):
        _same(b.A, a.A)
    L = H0.levels[0]
    assert (L.PT.br, L.PT.bc, L.P.br, L.P.bc) == ((6, 6, 6, 6) if rot else (6, 3, 3, 6))
    _same(device_galerkin(L.PT, L.A, L.P), H0.levels[1].A)


def test_galerkin_product_and_hierarchy_are_identical_with_the_device_hook():
    prob = fem.poisson_fast((30, 28, 26), dirichlet="right|top", jitter=0.2, seed=3)
    A = Matrix(prob.n, prob.n, 1, 1, prob.rowptr, prob.col, prob.val)
    try:
        assert _lib.device_setup(False) is False
        H0 = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=50, spw=1)
        assert _lib.device_setup(True, min_rows=0) is True
        H1 = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=50, spw=1)
    finally:
        _lib._device_setup = None          # back to the environment's default for the tests that follow
        _lib.device_setup()
    assert len(H0.levels) == len(H1.levels) >= 4
    for a, b in zip(H0.levels, H1.levels):
        _same(b.A, a.A)
        if a.P is not None:
            _same(b.P, a.P)
    L = H0.levels[0]
    _same(device_galerkin(L.PT, L.A, L.P), H0.levels[1].A)
    # and against scipy (another summation order: tolerance)
    Ac = (L.PT.to_scipy() @ L.A.to_scipy() @ L.P.to_scipy()).tocsr()
    assert abs(Ac - H1.levels[1].A.to_scipy()).max() < 1e-12 * abs(Ac).max()
