"""Every device format / kernel variant on synthetic matrices, through the C ABI, against scipy:
SELL-G for G = 1..16 (with and without 16-bit column deltas), the CSR-vector fallback for irregular rows, block CSR
3x3 / 2x2, the row-per-lane 6x6 kernel for W = 1, 2, 4 and rectangular transfer blocks (3x6, 6x3, 1x3, 6x6)."""
import numpy as np
import pytest
import scipy.sparse as sp

from ngsamg_amd import Matrix
from ngsamg_amd.hierarchy import Level

pytestmark = pytest.mark.gpu


class _H:
    def __init__(self, levels):
        self.levels = levels
        self.coarse_n = 0
        self.coarse_inv = np.empty(0)
        self.n_levels = len(levels)


def _level(A, P=None, PT=None):
    n, bs = A.n_rows, A.br
    return Level(A=A, P=P, PT=PT, free=np.ones(n, dtype=np.uint8), dinv=np.tile(np.eye(bs).reshape(-1), n),
                 coords=None, color=np.full(n, -1, dtype=np.int32), n_colors=0, agg=None)


def _rand_bcsr(rng, n_rows, n_cols, br, bc, row_len, spread=None, sort=True):
    """block CSR with `row_len(i)` blocks in row i; columns near the diagonal (spread) or anywhere"""
    rowptr = [0]
    cols = []
    for i in range(n_rows):
        L = int(row_len(i))
        L = max(0, min(L, n_cols))
        if spread is None:
            c = rng.choice(n_cols, size=L, replace=False)
        else:
            lo = max(0, min(n_cols - 1, int(i * n_cols / n_rows)) - spread)
            hi = min(n_cols, lo + 2 * spread + 1)
            c = lo + rng.choice(hi - lo, size=min(L, hi - lo), replace=False)
        cols.append(np.sort(c))
        rowptr.append(rowptr[-1] + len(c))
    col = np.concatenate(cols).astype(np.int32) if cols else np.empty(0, dtype=np.int32)
    val = rng.standard_normal((len(col), br, bc))
    return Matrix(n_rows, n_cols, br, bc, np.array(rowptr, dtype=np.int64), col, val)


def _dev(levels):
    from ngsamg_amd.device import DeviceAMGMatrix
    return DeviceAMGMatrix(_H(levels), sm_type="jacobi", clev="none", device=0)


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("n,avg,spread,expect_fmt", [
    (5000, 3, 40, "sell"), (5000, 9, 60, "sell"), (3000, 20, 100, "sell"), (2000, 45, 200, "sell"), (1500, 90, 400, "sell"),
    (4000, 12, None, "sell"),            # random columns: 16-bit deltas impossible -> 32-bit slices
    (3000, 0, 50, "csrvec"),             # strongly irregular row lengths -> padding too high -> CSR-vector
])
def test_scalar_formats(n, avg, spread, expect_fmt):
    rng = np.random.default_rng(n + avg)
    if avg == 0:
        rl = lambda i: 1 if i % 7 else 120
    else:
        rl = lambda i: avg + (i % 3) - 1
    A = _rand_bcsr(rng, n, n, 1, 1, rl, spread)
    dev = _dev([_level(A)])
    info = dev.matrix_info(0, "A")
    assert info["fmt"] == expect_fmt, info
    S = A.to_scipy()
    for k in range(2):
        x = rng.standard_normal(n)
        y = np.empty(n)
        dev.MatVec(0, x, y)
        assert _rel(y, S @ x) < 1e-13
    # Jacobi stage kernels on the same matrix (EP_PRE on the scaled image, EP_JAC, EP_RES)
    b = rng.standard_normal(n)
    x, r = np.empty(n), np.empty(n)
    dev.JacobiPre(0, b, x, r)
    assert _rel(x, 0.9 * b) < 1e-14 and _rel(r, b - S @ (0.9 * b)) < 1e-12
    xo = np.empty(n)
    dev.JacobiPost(0, x, b, xo)
    assert _rel(xo, x + 0.9 * (b - S @ x)) < 1e-12


@pytest.mark.parametrize("bs,avg,uniform", [(2, 9, False), (3, 15, False), (3, 40, False), (6, 8, False), (6, 30, False), (6, 70, False),
                                            (2, 9, True), (3, 15, True), (6, 15, True), (6, 44, True)])
def test_block_matvec_and_jacobi(bs, avg, uniform):
    """uniform row lengths -> block SELL (one lane per scalar row); varying lengths -> CSR block kernels"""
    rng = np.random.default_rng(bs * 100 + avg)
    n = 700
    A = _rand_bcsr(rng, n, n, bs, bs, (lambda i: avg) if uniform else (lambda i: max(1, avg + (i % 5) * (avg // 3) - (2 * avg) // 3)), 150)
    lev = _level(A)
    dinv = rng.standard_normal((n, bs, bs))
    lev.dinv = np.ascontiguousarray(dinv.reshape(-1))
    dev = _dev([lev])
    if uniform:
        assert dev.matrix_info(0, "A")["fmt"] == "bsell"
    S = A.to_scipy()
    x = rng.standard_normal(n * bs)
    y = np.empty(n * bs)
    dev.MatVec(0, x, y)
    assert _rel(y, S @ x) < 1e-13
    b = rng.standard_normal(n * bs)
    xo = np.empty(n * bs)
    dev.JacobiPost(0, x, b, xo)
    t = (b - S @ x).reshape(n, bs)
    ref = x + 0.9 * np.einsum("nij,nj->ni", dinv, t).reshape(-1)
    assert _rel(xo, ref) < 1e-12
    r = np.empty(n * bs)
    dev.Residual(0, x, b, r)
    assert _rel(r, b - S @ x) < 1e-13


@pytest.mark.parametrize("bf,bc_,avg", [(1, 1, 3), (3, 6, 4), (6, 6, 4), (2, 3, 3), (1, 3, 2), (1, 6, 3)])
def test_transfer_block_shapes(bf, bc_, avg):
    """P with bf x bc blocks, P^T with bc x bf blocks (reference ProlMap block shapes, dof_map.hpp:444-458)"""
    rng = np.random.default_rng(bf * 10 + bc_)
    nf, nc = 900, 120
    P = _rand_bcsr(rng, nf, nc, bf, bc_, lambda i: avg, 8)
    PTs = sp.bsr_matrix(P.to_scipy().T.tocsr(), blocksize=(bc_, bf))
    PTs.sort_indices()
    PT = Matrix(nc, nf, bc_, bf, PTs.indptr, PTs.indices, PTs.data)
    Af = _rand_bcsr(rng, nf, nf, bf, bf, lambda i: 3, 5)
    Ac = _rand_bcsr(rng, nc, nc, bc_, bc_, lambda i: 3, 5)
    dev = _dev([_level(Af, P, PT), _level(Ac)])
    Ps = P.to_scipy()
    xf = rng.standard_normal(nf * bf)
    xc = np.empty(nc * bc_)
    dev.TransferF2C(0, xf, xc)
    assert _rel(xc, Ps.T @ xf) < 1e-13
    xc = rng.standard_normal(nc * bc_)
    a = xf.copy()
    dev.AddC2F(0, -0.3, a, xc)
    assert _rel(a, xf - 0.3 * (Ps @ xc)) < 1e-13
    out = np.empty_like(xf)
    dev.Prolong(0, 1.0, xf, xc, out)
    assert _rel(out, xf + Ps @ xc) < 1e-13


@pytest.mark.parametrize("shape,rot", [((9, 8, 7), False), ((9, 8, 7), True), ((20, 17), False), ((20, 17), True)])
def test_rigid_body_transfer_blocks(shape, rot, monkeypatch):
    """P_ik = w_ik Q(t_ik) (elasticity_energy.hpp:447-490) stored as (column, w, t) instead of the bf x bc block:
    amgx_create detects the structure; TransferF2C / AddC2F (dof_map.cpp:636-709) agree with the general block kernels and
    with scipy on the explicit matrices, and a perturbed block falls back to the general format"""
    import scipy.sparse as sp
    from ngsamg_amd.device import DeviceAMGMatrix
    from tests.problems import elasticity_case
    p, H = elasticity_case(shape, rotations=rot, max_coarse_size=5)
    dev = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    monkeypatch.setenv("AMGX_NO_RB_TRANSFER", "1")
    gen = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    monkeypatch.delenv("AMGX_NO_RB_TRANSFER")
    rng = np.random.default_rng(0)
    for l in range(H.n_levels - 1):
        assert dev.matrix_info(l, "P")["fmt"] == "rigid-body" and dev.matrix_info(l, "PT")["fmt"] == "rigid-body"
        assert gen.matrix_info(l, "P")["fmt"] != "rigid-body"
        assert dev.matrix_info(l, "P")["stream_bytes"] < (0.3 if len(shape) == 3 else 0.7) * gen.matrix_info(l, "P")["stream_bytes"]
        P = H.levels[l].P.to_scipy()
        nf, nc = P.shape
        xf, xc = rng.standard_normal(nf), rng.standard_normal(nc)
        got, ref = np.empty(nc), np.empty(nc)
        dev.TransferF2C(l, xf, got)
        gen.TransferF2C(l, xf, ref)
        assert np.linalg.norm(got - P.T @ xf) <= 1e-13 * np.linalg.norm(P.T @ xf) and np.linalg.norm(got - ref) <= 1e-13 * np.linalg.norm(ref)
        y1, y2 = xf.copy(), xf.copy()
        dev.AddC2F(l, -0.7, y1, xc)
        gen.AddC2F(l, -0.7, y2, xc)
        assert np.linalg.norm(y1 - (xf - 0.7 * (P @ xc))) <= 1e-13 * np.linalg.norm(y1) and np.linalg.norm(y1 - y2) <= 1e-13 * np.linalg.norm(y2)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    x1, x2 = np.empty_like(b), np.empty_like(b)
    dev.Mult(b, x1)
    gen.Mult(b, x2)
    assert np.linalg.norm(x1 - x2) <= 1e-12 * np.linalg.norm(x2)
    # one block that is not a rigid-body transformation: the level keeps the general block format
    import copy
    H2 = _H(list(H.levels))
    H2.coarse_n, H2.coarse_inv = H.coarse_n, H.coarse_inv
    L0 = copy.copy(H.levels[0])
    Pm = copy.copy(L0.P)
    Pm.val = L0.P.val.copy()
    Pm.val.reshape(-1)[1] += 1e-3
    L0.P = Pm
    H2.levels[0] = L0
    dev2 = DeviceAMGMatrix(H2, sm_type="jacobi", device=0)
    assert dev2.matrix_info(0, "P")["fmt"] != "rigid-body" and dev2.matrix_info(0, "PT")["fmt"] == "rigid-body"
