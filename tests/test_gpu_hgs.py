"""Block-hybrid Gauss-Seidel (amgx_level_desc.gs_block_rows, gsb_sweep_kernel) against the oracle's serial hybrid GS with
the same blocks, colours and modified diagonal: cycles, smoother flag contract, and the iteration-count tie to the
reference's sequential Gauss-Seidel (within +15 %)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(shape, dirichlet="right|top", mcs=20):
    from tests.problems import poisson_case
    return poisson_case(shape, dirichlet, mcs)


@pytest.mark.parametrize("shape", [(21, 21, 21), (33, 30, 28), (70, 70)])
@pytest.mark.parametrize("cycle", ["V", "W", "BS"])
def test_hgs_cycles_match_hybrid_oracle(shape, cycle):
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    p, H = _case(shape)
    dev = DeviceAMGMatrix(H, sm_type="hgs", mg_cycle=cycle, device=0)
    assert any(h is not None for h in dev.hgs), "no level took the block-hybrid path"
    lv, types = hgs_levels(H.levels, dev.hgs)
    orc = Oracle(lv, sm_type=types, cycle=cycle)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    for rep in range(2):
        x = dev.apply(b)
    ref = orc.apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("shape,cap", [((40, 38, 30), None), ((40, 38, 30), 1500)])
def test_hgs_local_window_residual_on_long_row_levels(shape, cap, monkeypatch):
    """the residual after the sweep from zero + chunk-local restriction through the local-window image of the rest part
    (sell_lw_pre_restrict_kernel, MODE 1: the swept x staged in LDS), forced onto the small coarse levels of this case;
    cap: some chunks without a window (32-bit global columns)"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    monkeypatch.setenv("AMGX_LW_MIN_ROWS", "300")
    monkeypatch.setenv("AMGX_NO_DENSE_TAIL", "1")
    monkeypatch.setenv("AMGX_GSB_LW", "1")              # (the local-window form of the general sweep too: opt-in, a measured non-win)
    if cap:
        monkeypatch.setenv("AMGX_LW_TEST_CAP", str(cap))
    p, H = _case(shape)
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    lv, types = hgs_levels(H.levels, dev.hgs)
    rng = np.random.default_rng(2)
    b = rng.standard_normal(p.n) * p.free
    for rep in range(2):
        x = dev.apply(b)
    ref = Oracle(lv, sm_type=types).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    monkeypatch.setenv("AMGX_NO_LW", "1")
    xp = DeviceAMGMatrix(H, sm_type="hgs", device=0).apply(b)
    assert np.linalg.norm(x - xp) <= 1e-12 * np.linalg.norm(xp) and not np.array_equal(x, xp)


@pytest.mark.parametrize("threads", ["512", "1024"])
@pytest.mark.parametrize("split", [True, False])
def test_hgs_variants(threads, split, monkeypatch):
    """512-thread workgroups (other block sizes / lanes per row) and the pre-smoothing without the lower / rest split"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    monkeypatch.setenv("AMGX_GSB_THREADS", threads)
    if not split:
        monkeypatch.setenv("AMGX_GSB_NO_SPLIT", "1")
    p, H = _case((26, 24, 22))
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    assert dev.hgs[0]["B"] == int(threads)
    lv, types = hgs_levels(H.levels, dev.hgs)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(p.n) * p.free
    x = dev.apply(b)
    ref = Oracle(lv, sm_type=types).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("ru,ur,xz", [(False, False, False), (False, True, False), (True, True, True), (False, False, True)])
@pytest.mark.parametrize("back", [False, True])
def test_hgs_smoother_flag_contract(ru, ur, xz, back):
    """GetSmoother(level).Smooth / SmoothBack through amgx_smooth on a block-hybrid level (generic, out-of-place sweep)"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    p, H = _case((21, 21, 21))
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    lv, types = hgs_levels(H.levels, dev.hgs)
    orc = Oracle(lv, sm_type=types)
    rng = np.random.default_rng(2)
    A0 = H.levels[0].A.to_scipy()
    b = rng.standard_normal(p.n) * p.free
    x0 = np.zeros(p.n) if xz else rng.standard_normal(p.n) * p.free
    r0 = b - A0 @ x0 if ru else rng.standard_normal(p.n)
    xg, rg = x0.copy(), r0.copy()
    dev.Smooth(0, xg, b, rg, ru, ur, xz, back=back)
    xo, ro = orc.smooth(0, x0.copy(), b, r0.copy(), ru, ur, xz, back)
    assert np.linalg.norm(xg - xo) <= 1e-11 * max(1.0, np.linalg.norm(xo))
    if ur:
        fr = p.free.astype(bool)
        assert np.linalg.norm((rg - ro)[fr]) <= 1e-10 * max(1.0, np.linalg.norm(ro[fr]))


def test_hgs_iterations_within_15_percent_of_sequential_gs():
    """PCG with the block-hybrid smoother vs the reference's sequential Gauss-Seidel (oracle 'gs') on cfg 1's shape"""
    import torch
    from ngsamg_amd.device import DeviceAMGMatrix
    from ngsamg_amd.krylov import CGSolver
    from oracle.pyoracle import Oracle
    p, H = _case((101, 101), "left|top", 5)
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    cg = CGSolver(dev, dev, tol=1e-12, maxsteps=200)
    cg.Solve(torch.from_numpy(b).cuda())
    _, it_seq, _ = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-12, maxit=200)
    assert cg.iterations <= int(np.ceil(1.15 * it_seq)), (cg.iterations, it_seq)


@pytest.mark.parametrize("rot", [False, True])
@pytest.mark.parametrize("cycle,split", [("V", True), ("W", True), ("V", False)])
def test_block_gs_on_colour_major_bsell(rot, cycle, split, monkeypatch):
    """point-block Gauss-Seidel of the elasticity levels through bgs_bsell_color_kernel (colour-major BSELL copy; big
    levels only in production, forced onto the small test levels here) == the oracle's GS in colour order.  split: the
    pre-smoothing from zero reads the lower-colour couplings for the sweep and the higher-colour ones for the residual"""
    from tests.problems import elasticity_case
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_BGS_BSELL_MIN", "1")
    if not split:
        monkeypatch.setenv("AMGX_NO_BGS_SPLIT", "1")
    p, H = elasticity_case((9, 8, 7), rot, 10)
    dev = DeviceAMGMatrix(H, sm_type="gs", mg_cycle=cycle, device=0)
    rng = np.random.default_rng(3)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    for rep in range(2):
        x = dev.apply(b)
    ref = Oracle(H.levels, sm_type="gs_mc", cycle=cycle).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


def test_in_cycle_kernel_probes():
    """amgx_time_op op 8 / op 9: HIP events around the dominant kernel while whole cycles run (what bench.py reports as
    roofline.kernel_ms); the probe must not change the result of later applications"""
    import torch
    from ngsamg_amd.device import DeviceAMGMatrix
    from ngsamg_amd._lib import NgsAMGError
    p, H = _case((40, 40, 40), mcs=30)
    rng = np.random.default_rng(1)
    b = torch.from_numpy(rng.standard_normal(p.n) * p.free).cuda()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for sm, op, other in (("hgs", 9, 8), ("jacobi", 8, 9)):
            dev = DeviceAMGMatrix(H, sm_type=sm, device=0)
            x0 = torch.empty_like(b); dev.Mult(b, x0)
            s.synchronize()
            try:
                t = dev.time_op(0, op, reps=3)
                assert 0.0 < t < 50.0
            except NgsAMGError as e:            # (the fused Jacobi kernel exists on one-thread-per-row levels only: >= 2^20 rows)
                assert sm == "jacobi" and "no fused" in str(e)
            with pytest.raises(NgsAMGError):
                dev.time_op(0, other, reps=2)
            x1 = torch.empty_like(b); dev.Mult(b, x1)
            s.synchronize()
            assert torch.equal(x0, x1)


# ---- square-block levels: bgsb_sweep_kernel (reference HybridGSSmoother<Mat<3,3>> / <Mat<6,6>>, gssmoother.cpp:891-896) ----

@pytest.mark.parametrize("rot,shape", [(False, (14, 13, 12)), (True, (12, 11, 10)), (False, (40, 36))])
@pytest.mark.parametrize("cycle,split", [("V", True), ("W", True), ("V", False)])
def test_block_levels_hgs_cycles_match_hybrid_oracle(rot, shape, cycle, split, monkeypatch):
    """elasticity levels (3x3 fine / 6x6 coarse, 6x6 everywhere, 2x2 / 3x3 in 2D) in the block-hybrid form: one launch per
    sweep, blocks of ~128 block rows, l1-modified block diagonal; against the oracle's serial hybrid sweep"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    from tests.hgs_oracle import hgs_levels
    if not split:         # pre-smoothing as sweep + full residual instead of the one-pass lower / rest copies
        monkeypatch.setenv("AMGX_BGSB_NO_SPLIT", "1")
    p, H = elasticity_case(shape, rotations=rot, max_coarse_size=10)
    dev = DeviceAMGMatrix(H, sm_type="hgs", mg_cycle=cycle, device=0)
    assert dev.hgs[0] is not None and dev.hgs[0]["B"] % (64 // H.levels[0].bs) == 0, "level 0 did not take the block-hybrid path"
    lv, types = hgs_levels(H.levels, dev.hgs)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    for rep in range(2):
        x = dev.apply(b)
    ref = Oracle(lv, sm_type=types, cycle=cycle).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("rot,shape", [(False, (14, 13, 12)), (True, (12, 11, 10)), (False, (40, 36))])
@pytest.mark.parametrize("cycle,split", [("V", True), ("W", True), ("V", False)])
def test_block_levels_block_coloured_gs_matches_ordered_oracle(rot, shape, cycle, split, monkeypatch):
    """amgx_level_desc.gs_block_color (the default on square-block levels with >= 50 k block rows, forced onto these small ones):
    the sweep blocks carry a colouring of the block graph, a sweep is one in-place launch per block colour = exact Gauss-Seidel in the
    order (block colour, block, in-block colour); the oracle runs GSS3's loop (gssmoother.cpp:196-257) in exactly that order"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    from tests.hgs_oracle import hgs_levels
    monkeypatch.setenv("AMGX_BGSB_BC_MIN_ROWS", "0")
    if not split:
        monkeypatch.setenv("AMGX_BGSB_NO_SPLIT", "1")
    p, H = elasticity_case(shape, rotations=rot, max_coarse_size=10)
    dev = DeviceAMGMatrix(H, sm_type="hgs", mg_cycle=cycle, device=0)
    assert dev.hgs[0] is not None and dev.hgs[0]["block_color"] is not None and dev.hgs[0]["n_block_colors"] >= 2
    lv, types = hgs_levels(H.levels, dev.hgs)
    assert lv[0].gs_block is None
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    for rep in range(2):
        x = dev.apply(b)
    ref = Oracle(lv, sm_type=types, cycle=cycle).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("rot", [False, True])
def test_block_coloured_gs_flags_symmetry_and_iteration_tolerance(rot, monkeypatch):
    """flag contract of the in-place block-coloured sweeps, symmetry of the cycle, and SURVEY 8d's tolerance for a GPU-parallel
    Gauss-Seidel order: PCG iterations within +15 % of the reference's sequential order"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    from tests.hgs_oracle import hgs_levels
    monkeypatch.setenv("AMGX_BGSB_BC_MIN_ROWS", "0")
    p, H = elasticity_case((16, 14, 12), rotations=rot, max_coarse_size=10)
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    assert dev.hgs[0]["block_color"] is not None
    lv, types = hgs_levels(H.levels, dev.hgs)
    orc = Oracle(lv, sm_type=types)
    rng = np.random.default_rng(5)
    n = p.n * p.bs
    fr = np.repeat(p.free, p.bs).astype(np.float64)
    A = H.levels[0].A.to_scipy()
    for back in (False, True):
        for ru, ur, xz in ((False, False, False), (False, True, False), (True, True, True), (False, False, True)):
            b = rng.standard_normal(n) * fr
            x = np.zeros(n) if xz else rng.standard_normal(n) * fr
            res = (b - A @ x) if ru else rng.standard_normal(n)
            xo, ro = orc.smooth(0, x.copy(), b, res.copy(), ru, ur, xz, back)
            dev.Smooth(0, x, b, res, ru, ur, xz, back=back)
            assert np.linalg.norm(x - xo) <= 1e-10 * max(np.linalg.norm(xo), 1.0)
            if ur:
                f = fr.astype(bool)
                assert np.linalg.norm((res - ro)[f]) <= 1e-9 * max(np.linalg.norm(ro[f]), 1.0)
    u, v = rng.standard_normal(n) * fr, rng.standard_normal(n) * fr
    assert abs(np.dot(v, dev.apply(u)) - np.dot(u, dev.apply(v))) <= 1e-10 * abs(np.dot(v, dev.apply(u)))
    b = rng.standard_normal(n) * fr
    it_h = orc.pcg(b, tol=1e-8, maxit=200)[1]
    it_seq = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=200)[1]
    assert it_h <= int(np.ceil(1.15 * it_seq))


@pytest.mark.parametrize("rot", [False, True])
def test_block_levels_hgs_compact_sweep_blocks(rot, monkeypatch):
    """AMGX_BGSB_COMPACT=1: sweep blocks grown over the matrix graph (amgx_level_desc.gs_block_ids) instead of runs of
    consecutive rows -- same kernel, same oracle (blocks = the ids), fewer frozen couplings"""
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    from tests.hgs_oracle import hgs_levels
    monkeypatch.setenv("AMGX_BGSB_COMPACT", "1")
    p, H = elasticity_case((14, 13, 12), rotations=rot, max_coarse_size=10)
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    assert dev.hgs[0] is not None and dev.hgs[0]["block_of_row"] is not None
    lv, types = hgs_levels(H.levels, dev.hgs)
    rng = np.random.default_rng(1)
    b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
    x = dev.apply(b)
    ref = Oracle(lv, sm_type=types).apply(b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    monkeypatch.delenv("AMGX_BGSB_COMPACT")
    lines = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    assert lines.hgs[0]["block_of_row"] is None
    lvl, typl = hgs_levels(H.levels, lines.hgs)
    it_c = Oracle(lv, sm_type=types).pcg(b, tol=1e-8, maxit=200)[1]
    it_l = Oracle(lvl, sm_type=typl).pcg(b, tol=1e-8, maxit=200)[1]
    assert it_c <= it_l


@pytest.mark.parametrize("rot", [False, True])
def test_block_levels_hgs_smoother_flags_and_iterations(rot):
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.problems import elasticity_case
    from tests.hgs_oracle import hgs_levels
    p, H = elasticity_case((16, 14, 12), rotations=rot, max_coarse_size=10)
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    lv, types = hgs_levels(H.levels, dev.hgs)
    orc = Oracle(lv, sm_type=types)
    rng = np.random.default_rng(3)
    n = p.n * p.bs
    fr = np.repeat(p.free, p.bs).astype(np.float64)
    A = H.levels[0].A.to_scipy()
    for back in (False, True):
        for ru, ur, xz in ((False, False, False), (False, True, False), (True, True, True), (False, False, True)):
            b = rng.standard_normal(n) * fr
            x = np.zeros(n) if xz else rng.standard_normal(n) * fr
            res = (b - A @ x) if ru else rng.standard_normal(n)
            xo, ro = orc.smooth(0, x.copy(), b, res.copy(), ru, ur, xz, back)
            dev.Smooth(0, x, b, res, ru, ur, xz, back=back)
            assert np.linalg.norm(x - xo) <= 1e-10 * max(np.linalg.norm(xo), 1.0)
            if ur:
                f = fr.astype(bool)
                assert np.linalg.norm((res - ro)[f]) <= 1e-9 * max(np.linalg.norm(ro[f]), 1.0)
    # symmetric preconditioner, and PCG stays inside SURVEY 8d's +15 % of the reference's sequential order (13 / 11 and 15 / 13 here)
    u, v = rng.standard_normal(n) * fr, rng.standard_normal(n) * fr
    assert abs(np.dot(v, dev.apply(u)) - np.dot(u, dev.apply(v))) <= 1e-10 * abs(np.dot(v, dev.apply(u)))
    b = rng.standard_normal(n) * fr
    it_h = orc.pcg(b, tol=1e-8, maxit=200)[1]
    it_seq = Oracle(H.levels, sm_type="gs").pcg(b, tol=1e-8, maxit=200)[1]
    assert it_h <= int(np.ceil(1.15 * it_seq)), (it_h, it_seq)
