"""GSS4 on the CPU side: the oracle restatement (oracle/gss4.c, reference gssmoother.cpp:407-583) against an independent
numpy transcription of the same loops, and its structural properties.  No GPU needed."""
import numpy as np
import pytest
import scipy.sparse as sp


def _problem(kind, seed=0):
    from ngsamg_amd import fem
    from tests.problems import to_matrix
    if kind == "poisson":
        p = fem.poisson_fast((9, 8, 7))
    elif kind == "elast3":
        p = fem.elasticity_fast((6, 5, 4), dirichlet="left", mu=1.0, lam=0.5)
    else:
        p = fem.elasticity_fast((5, 4, 4), dirichlet="left", mu=1.0, lam=0.5, rotations=True)
    A = to_matrix(p)
    rng = np.random.default_rng(seed)
    subset = (rng.random(p.n) < 0.3) & (p.free > 0)
    return p, A, subset.astype(np.uint8), rng


def _dinv(A, subset, repl=None):
    bs = A.br
    D = np.zeros((A.n_rows, bs, bs))
    S = A.to_scipy().tocsr()
    for k in np.flatnonzero(subset):
        blk = S[k * bs:(k + 1) * bs, k * bs:(k + 1) * bs].toarray() if repl is None else repl[k]
        D[k] = np.linalg.inv(blk)
    return D.ravel()


def _numpy_gss4(A, subset, dinv, x, v, form, back):
    """transcription of SmoothRHSInternal / SmoothRESInternal with dense row access (independent of gss4.c)"""
    bs = A.br
    S = A.to_scipy().tocsr()
    xd = np.flatnonzero(subset)
    D = dinv.reshape(-1, bs, bs)
    x = x.copy(); v = v.copy()
    for k in (xd[::-1] if back else xd):
        rows = slice(k * bs, (k + 1) * bs)
        Ak = S[rows, :]
        if form == "rhs":
            x[rows] += D[k] @ (v[rows] - Ak @ x)
        else:
            w = -D[k] @ v[rows]
            v += Ak.T @ w
            x[rows] -= w
    return x, v


@pytest.mark.parametrize("kind", ["poisson", "elast3", "elast6"])
@pytest.mark.parametrize("back", [False, True])
def test_oracle_gss4_matches_numpy_transcription(kind, back):
    from oracle.pyoracle import OracleGSS4
    p, A, subset, rng = _problem(kind)
    dinv = _dinv(A, subset)
    g = OracleGSS4(A, subset, dinv)
    assert g.rows == int(subset.sum())
    n = p.n * p.bs
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    x = x0.copy(); g.smooth(x, b, back)
    ref, _ = _numpy_gss4(A, subset, dinv, x0, b, "rhs", back)
    assert np.allclose(x, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    assert np.array_equal(x[np.repeat(subset == 0, p.bs)], x0[np.repeat(subset == 0, p.bs)])      # only the subset moves
    x = x0.copy(); r = b.copy(); g.smooth_res(x, r, back)
    refx, refr = _numpy_gss4(A, subset, dinv, x0, b, "res", back)
    assert np.allclose(x, refx, rtol=0, atol=1e-13 * np.abs(refx).max())
    assert np.allclose(r, refr, rtol=0, atol=1e-13 * np.abs(refr).max())


def test_oracle_gss4_res_form_keeps_the_residual_and_equals_gss3_on_the_subset():
    """symmetric A: RES form == RHS form with res = b - A x kept up to date; GSS4 == GSS3 whose free mask is the subset"""
    from oracle.pyoracle import OracleGSS4, Oracle
    from tests.problems import poisson_case
    p, H = poisson_case((9, 8, 7))
    rng = np.random.default_rng(5)
    A = H.levels[0].A
    S = A.to_scipy().tocsr()
    subset = ((rng.random(p.n) < 0.4) & (p.free > 0)).astype(np.uint8)
    dinv = _dinv(A, subset)
    g = OracleGSS4(A, subset, dinv)
    x0, b = rng.standard_normal(p.n), rng.standard_normal(p.n)
    for back in (False, True):
        xa = x0.copy(); g.smooth(xa, b, back)
        xb = x0.copy(); r = b - S @ x0; g.smooth_res(xb, r, back)
        assert np.allclose(xa, xb, rtol=0, atol=1e-12)
        assert np.allclose(r, b - S @ xb, rtol=0, atol=1e-11)
    # GSS3 with free = subset and the same inverse diagonal
    import copy
    lv = copy.copy(H.levels[0])
    lv.free = subset
    lv.dinv = dinv
    orc = Oracle([lv], sm_type="gs", clev="none")
    xa = x0.copy(); g.smooth(xa, b, False)
    xc = x0.copy(); res = np.zeros_like(b); orc.smooth(0, xc, b, res, back=False)
    assert np.allclose(xa, xc, rtol=0, atol=1e-13)


def test_oracle_gss4_mult_add_and_visiting_order():
    from oracle.pyoracle import OracleGSS4
    p, A, subset, rng = _problem("elast3")
    dinv = _dinv(A, subset)
    g = OracleGSS4(A, subset, dinv)
    n = p.n * p.bs
    b, x0 = rng.standard_normal(n), rng.standard_normal(n)
    x = x0.copy(); g.mult_add(0.7, b, x)
    D = dinv.reshape(-1, 3, 3)
    ref = x0.copy().reshape(-1, 3)
    idx = np.flatnonzero(subset)
    ref[idx] += 0.7 * np.einsum("kij,kj->ki", D[idx], b.reshape(-1, 3)[idx])
    assert np.allclose(x, ref.ravel(), rtol=0, atol=1e-14)
    # a reversed visiting order forward == natural order backward
    m = g.rows
    g2 = OracleGSS4(A, subset, dinv, order=np.arange(m - 1, -1, -1))
    xa = x0.copy(); g.smooth(xa, b, True)
    xb = x0.copy(); g2.smooth(xb, b, False)
    assert np.array_equal(xa, xb)
    with pytest.raises(RuntimeError):
        OracleGSS4(A, subset, dinv, order=np.arange(m - 1))
