"""GSS4 on the GPU (amgx_gss4_*, ngsamg_amd.NgsAMG.GSS4) against the CPU restatement of the reference's GSS4
(oracle/gss4.c, gssmoother.cpp:407-583).  The GPU relaxes the rows of one colour in parallel, so the oracle visits the
compressed rows in colour-major order; on an independent subset (no two rows coupled) every order gives the reference's
ascending-order result, which is checked too."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(kind, seed=0, frac=0.3):
    from ngsamg_amd import fem
    from tests.problems import to_matrix
    if kind == "poisson":
        p = fem.poisson_fast((17, 15, 13))
    elif kind == "poisson2d":
        p = fem.poisson_fast((40, 37))
    elif kind == "elast3":
        p = fem.elasticity_fast((9, 8, 7), dirichlet="left", mu=1.0, lam=0.5)
    else:
        p = fem.elasticity_fast((7, 6, 5), dirichlet="left", mu=1.0, lam=0.5, rotations=True)
    rng = np.random.default_rng(seed)
    subset = ((rng.random(p.n) < frac) & (p.free > 0)).astype(np.uint8)
    return p, to_matrix(p), subset, rng


def _colour_order(g):
    xd = np.flatnonzero(g.subset)
    return np.argsort(g.color[xd], kind="stable").astype(np.int32)


def _check(a, ref, tol=1e-12):
    assert np.linalg.norm(a - ref) <= tol * max(1.0, np.linalg.norm(ref))


@pytest.mark.parametrize("kind", ["poisson", "poisson2d", "elast3", "elast6"])
def test_gss4_all_forms_match_oracle(kind):
    from ngsamg_amd.NgsAMG import GSS4
    from oracle.pyoracle import OracleGSS4
    p, A, subset, rng = _problem(kind)
    g = GSS4(A, subset, bs=p.bs)
    info = g.info()
    assert info["rows"] == int(subset.sum()) and info["rows_touched"] >= info["rows"] and info["colors"] >= 1
    orc = OracleGSS4(A, subset, g.dinv, order=_colour_order(g))
    assert orc.rows == info["rows"] and orc.nnz == info["nnz"]
    n = p.n * p.bs
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    mask = np.repeat(subset == 0, p.bs)
    for back in (False, True):
        x = x0.copy(); (g.SmoothBack if back else g.Smooth)(x, b)
        ref = x0.copy(); orc.smooth(ref, b, back)
        _check(x, ref)
        assert np.array_equal(x[mask], x0[mask])
        x = x0.copy(); r = b.copy(); (g.SmoothBackRES if back else g.SmoothRES)(x, r)
        refx = x0.copy(); refr = b.copy(); orc.smooth_res(refx, refr, back)
        _check(x, refx)
        _check(r, refr)
        assert np.array_equal(x[mask], x0[mask])
    x = x0.copy(); g.MultAdd(-0.35, b, x)
    ref = x0.copy(); orc.mult_add(-0.35, b, ref)
    _check(x, ref, 1e-14)


def test_gss4_device_tensors_and_repeated_sweeps():
    import torch
    from ngsamg_amd.NgsAMG import GSS4
    from oracle.pyoracle import OracleGSS4
    p, A, subset, rng = _problem("poisson", seed=3, frac=0.5)
    g = GSS4(A, subset)
    orc = OracleGSS4(A, subset, g.dinv, order=_colour_order(g))
    x0, b = rng.standard_normal(p.n), rng.standard_normal(p.n)
    xd, bd = torch.from_numpy(x0).cuda(), torch.from_numpy(b).cuda()
    ref = x0.copy()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(3):
            g.Smooth(xd, bd); g.SmoothBack(xd, bd)
    s.synchronize()
    for k in range(3):
        orc.smooth(ref, b, False); orc.smooth(ref, b, True)
    _check(xd.cpu().numpy(), ref, 1e-11)
    # RES form on device tensors, residual stays consistent with b - A x (symmetric matrix)
    S = A.to_scipy().tocsr()
    x = torch.from_numpy(x0).cuda()
    r = torch.from_numpy(b - S @ x0).cuda()
    g.SmoothRES(x, r); g.SmoothBackRES(x, r)
    torch.cuda.synchronize()
    xh, rh = x.cpu().numpy(), r.cpu().numpy()
    _check(rh, b - S @ xh, 1e-11)
    with pytest.raises(Exception):
        g.Smooth(xd, b)            # mixing device and host vectors


def test_gss4_independent_subset_equals_reference_order():
    """no two rows of the subset are coupled: one colour, and the result equals the reference's one-row-at-a-time sweep"""
    from ngsamg_amd.NgsAMG import GSS4
    from oracle.pyoracle import OracleGSS4
    p, A, _, rng = _problem("poisson")
    S = A.to_scipy().tocsr()
    subset = np.zeros(p.n, dtype=np.uint8)
    blocked = np.zeros(p.n, dtype=bool)
    for k in rng.permutation(p.n):
        if p.free[k] and not blocked[k]:
            subset[k] = 1
            blocked[S.indices[S.indptr[k]:S.indptr[k + 1]]] = True
    g = GSS4(A, subset)
    assert g.n_colors == 1
    orc = OracleGSS4(A, subset, g.dinv)             # ascending order = the reference
    x0, b = rng.standard_normal(p.n), rng.standard_normal(p.n)
    for back in (False, True):
        x = x0.copy(); (g.SmoothBack if back else g.Smooth)(x, b)
        ref = x0.copy(); orc.smooth(ref, b, back)
        _check(x, ref, 1e-14)
        x = x0.copy(); r = b.copy(); (g.SmoothBackRES if back else g.SmoothRES)(x, r)
        refx, refr = x0.copy(), b.copy(); orc.smooth_res(refx, refr, back)
        _check(x, refx, 1e-14); _check(r, refr, 1e-14)


def test_gss4_replacement_diagonal_pinv_and_nonsymmetric_matrix():
    """repl_diag (the hybrid smoother's mod_diag) with and without pinv; the RES form is literal for a non-symmetric A too"""
    import scipy.sparse as sp
    from ngsamg_amd.NgsAMG import GSS4
    from ngsamg_amd._lib import Matrix
    from oracle.pyoracle import OracleGSS4
    p, A, subset, rng = _problem("elast3", seed=7)
    S = A.to_scipy().tocsr()
    bs = 3
    repl = np.zeros((p.n, bs, bs))
    for k in range(p.n):
        repl[k] = 1.3 * S[k * bs:(k + 1) * bs, k * bs:(k + 1) * bs].toarray() + 0.1 * np.eye(bs)
    sing = np.flatnonzero(subset)[:5]
    repl[sing, 2, :] = 0.0; repl[sing, :, 2] = 0.0             # singular replacement blocks: only pinv can take them
    g = GSS4(A, subset, repl_diag=repl, pinv=True, bs=bs)
    D = g.dinv.reshape(-1, bs, bs)
    assert np.allclose(D[sing[0]] @ repl[sing[0]] @ D[sing[0]], D[sing[0]], atol=1e-12)
    orc = OracleGSS4(A, subset, g.dinv, order=_colour_order(g))
    n = p.n * bs
    x0, b = rng.standard_normal(n), rng.standard_normal(n)
    x = x0.copy(); g.Smooth(x, b)
    ref = x0.copy(); orc.smooth(ref, b, False)
    _check(x, ref)
    # non-symmetric values on the symmetric pattern
    Sn = S.copy()
    Sn.data = Sn.data * (1.0 + 0.3 * rng.random(Sn.data.size))
    An = Matrix.from_scipy(sp.csr_matrix(Sn), bs)
    g2 = GSS4(An, subset, bs=bs)
    orc2 = OracleGSS4(An, subset, g2.dinv, order=_colour_order(g2))
    for back in (False, True):
        x = x0.copy(); r = b.copy(); (g2.SmoothBackRES if back else g2.SmoothRES)(x, r)
        refx, refr = x0.copy(), b.copy(); orc2.smooth_res(refx, refr, back)
        _check(x, refx); _check(r, refr)


def test_gss4_ghost_columns_empty_subset_and_errors():
    """rectangular matrix (owned rows x [owned | ghost]): ghost entries are read, never written; empty subset is a no-op;
    a colouring with two coupled rows of one colour is rejected"""
    import ctypes as C
    import scipy.sparse as sp
    from ngsamg_amd import _lib
    from ngsamg_amd.NgsAMG import GSS4
    from ngsamg_amd._lib import Matrix
    from oracle.pyoracle import OracleGSS4
    p, A, subset, rng = _problem("poisson", seed=11)
    S = A.to_scipy().tocsr()
    n_own = p.n - 300
    R = sp.csr_matrix(S[:n_own, :])
    Ar = Matrix.from_scipy(R, 1)
    assert Ar.n_cols == p.n and Ar.n_rows == n_own
    sub = subset[:n_own].copy()
    g = GSS4(Ar, sub)
    orc = OracleGSS4(Ar, sub, g.dinv, order=_colour_order(g))
    x0, b = rng.standard_normal(p.n), rng.standard_normal(n_own)
    x = x0.copy(); g.Smooth(x, b)
    ref = x0.copy(); orc.smooth(ref, b, False)
    _check(x, ref)
    assert np.array_equal(x[n_own:], x0[n_own:])
    xs, r = rng.standard_normal(n_own), rng.standard_normal(p.n)
    xa, ra = xs.copy(), r.copy(); g.SmoothBackRES(xa, ra)
    xb, rb = xs.copy(), r.copy(); orc.smooth_res(xb, rb, True)
    _check(xa, xb); _check(ra, rb)
    # empty subset
    g0 = GSS4(A, np.zeros(p.n, dtype=np.uint8))
    x = x0.copy(); g0.Smooth(x, x0 * 2); g0.SmoothRES(x, x0 * 3); g0.MultAdd(1.0, x0, x)
    assert np.array_equal(x, x0) and g0.info()["rows"] == 0
    # invalid colouring through the C ABI
    lib = _lib.hip()
    d = _lib.amgx_gss4_desc()
    d.A = A.desc(_lib.amgx_matrix)
    ones = np.ones(p.n, dtype=np.uint8)
    col = np.zeros(p.n, dtype=np.int32)
    dinv = np.ones(p.n)
    d.subset, d.dinv, d.color = _lib.ptr(ones, C.c_uint8), _lib.ptr(dinv, C.c_double), _lib.ptr(col, C.c_int32)
    d.n_colors, d.device = 1, 0
    h = C.c_void_p()
    assert lib.amgx_gss4_create(C.byref(d), C.byref(h)) != 0
    assert b"share a colour" in lib.amgx_gss4_last_error(None)
