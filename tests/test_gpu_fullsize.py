"""BASELINE.json's configurations at FULL size on one GPU: one application against the oracle (the oracle finishes a single
application in about a second on the box's cores) plus size-independent properties -- the V(1,1) cycle with Jacobi
smoothing is a symmetric operator (the reference checks the same asymmetry, amg_pc.cpp:162-173) and is linear."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _threads():
    import os
    n = os.cpu_count() or 8
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    return min(n, 32)


def _apply(dev, b):
    import torch
    bd = torch.from_numpy(b).cuda()
    xd = torch.empty_like(bd)
    dev.Mult(bd, xd)
    torch.cuda.synchronize()
    return xd.cpu().numpy()


def _properties(dev, n, free_s, rng):
    u = rng.standard_normal(n) * free_s
    v = rng.standard_normal(n) * free_s
    Cu, Cv = _apply(dev, u), _apply(dev, v)
    assert abs(np.dot(v, Cu) - np.dot(u, Cv)) <= 1e-10 * abs(np.dot(v, Cu))            # symmetric preconditioner
    Cw = _apply(dev, 2.0 * u - 3.0 * v)
    assert np.linalg.norm(Cw - (2.0 * Cu - 3.0 * Cv)) <= 1e-12 * np.linalg.norm(Cw)    # linear


def test_cfg1_2d_poisson_50k_gauss_seidel_iterations():
    """cfg 1: 2D H1 Poisson, 224^2 = 50 176 DOF, default (Gauss-Seidel) smoother, PCG to 1e-12, the reference's budget of
    30 iterations (tests/h1/test_2d_poisson.py, SURVEY.md 8d "expect < 30") with the default hierarchy (SPW agglomeration, semi-aux
    smoothed prolongation) -- met by the GPU's own sweep order, which needs at most 15 % more iterations than the reference's
    sequential order on the same hierarchy"""
    from ngsamg_amd import fem, ngs_amg, Matrix
    from ngsamg_amd.harness import Solve
    from oracle.pyoracle import Oracle
    p = fem.poisson_fast((224, 224), dirichlet="left|top")
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    c = ngs_amg.Preconditioner(A, "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2)
    sol, cg = Solve(c, p.load, ms=30, tol=1e-12, quiet=True)        # ms = 30: Solve asserts cg.iterations < 30, the reference's budget
    _, it_seq, _ = Oracle(c.GetHierarchy().levels, sm_type="gs", threads=_threads()).pcg(p.load, tol=1e-12, maxit=100)
    print("cfg 1 iterations: GPU (block-hybrid order)", cg.iterations, "sequential order", it_seq)
    assert it_seq < 30, it_seq                        # the same budget in the reference's (sequential) sweep order: 24
    assert cg.iterations <= int(np.ceil(1.15 * it_seq)), (cg.iterations, it_seq)
    f = p.free.astype(bool)
    assert np.linalg.norm((p.to_scipy() @ sol - p.load)[f]) < 1e-9 * np.linalg.norm(p.load)


def test_cfg2_3d_poisson_10m_jacobi_and_gauss_seidel():
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    from tests.hgs_oracle import hgs_levels
    p = fem.poisson_fast((215, 215, 215), dirichlet="right|top", jitter=0.2, seed=1)
    H = Hierarchy(Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val), p.free, p.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    dev = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    ref = Oracle(H.levels, sm_type="jacobi", threads=_threads()).apply(b)
    x = _apply(dev, b)
    assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref)
    _properties(dev, p.n, p.free.astype(np.float64), rng)
    del dev
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    lv, types = hgs_levels(H.levels, dev.hgs)
    ref = Oracle(lv, sm_type=types, threads=_threads()).apply(b)
    x = _apply(dev, b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("rot", [False, True])
def test_cfg3_cfg5_elasticity_126_cubed(rot):
    """cfg 3 (3x3 fine / 6x6 coarse blocks, 6.0 M DOF) and cfg 5 (6x6 blocks everywhere, 12.0 M DOF)"""
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle
    p = fem.elasticity_fast((126, 126, 126), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
    H = Hierarchy(Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val), p.free, p.coords, dim=3, energy=1, max_coarse_size=50,
                  regularize_cmats=0 if rot else 1)
    rng = np.random.default_rng(0)
    fs = np.repeat(p.free, p.bs).astype(np.float64)
    b = rng.standard_normal(p.n * p.bs) * fs
    dev = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
    ref = Oracle(H.levels, sm_type="jacobi", threads=_threads()).apply(b)
    x = _apply(dev, b)
    assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref)
    _properties(dev, p.n * p.bs, fs, rng)
    if rot:                       # cfg 5's convergent smoother: point-block Gauss-Seidel (colour-major BSELL kernel on level 0)
        del dev
        dev = DeviceAMGMatrix(H, sm_type="gs", device=0)
        ref = Oracle(H.levels, sm_type="gs_mc", threads=_threads()).apply(b)
        x = _apply(dev, b)
        assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    # the default Gauss-Seidel form since round 3: block-hybrid sweeps on the block levels (bgsb_sweep_kernel), one launch per sweep
    del dev
    from tests.hgs_oracle import hgs_levels
    dev = DeviceAMGMatrix(H, sm_type="hgs", device=0)
    assert dev.hgs[0] is not None and dev.hgs[0]["B"] in (120, 126)
    assert dev.hgs[0]["block_color"] is not None, "the big block levels sweep in the block-coloured form"
    lv, types = hgs_levels(H.levels, dev.hgs)
    ref = Oracle(lv, sm_type=types, threads=_threads()).apply(b)
    x = _apply(dev, b)
    assert np.linalg.norm(x - ref) <= 1e-10 * np.linalg.norm(ref)
    # a GPU-parallel Gauss-Seidel order may cost at most 15 % more PCG iterations than the reference's sequential sweep
    import torch
    from ngsamg_amd.krylov import CGSolver
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        cg = CGSolver(dev, dev, tol=1e-8, maxsteps=200)
        cg.Solve(torch.from_numpy(b).cuda())
        st.synchronize()
    it_seq = Oracle(H.levels, sm_type="gs", threads=_threads()).pcg(b, tol=1e-8, maxit=200)[1]
    assert cg.iterations <= int(np.ceil(1.15 * it_seq)), (cg.iterations, it_seq)


def test_cfg4_arrangement_8_virtual_ranks_production_formats():
    """cfg 4's arrangement (2 x 2 x 2 boxes, up to 7 neighbours per rank: faces, edges, corner) with 104^3 vertices per rank
    = 9 M DOF, eight virtual ranks on ONE GPU through the native driver (production formats: fused down kernel, windowed Q,
    interior / boundary launches, pack kernels), against the serial oracle on the assembled global hierarchy.  The full
    cfg 4 (171^3 per rank on 8 GPUs over RCCL) is `bench.py --config cfg4 --gpus 8` / tests/test_gpu_multi.py."""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    R, box = 8, (104, 104, 104)
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=20000, device=0, max_coarse_size=50, max_levels=10)
    assert amg.k >= 2 and amg.fold
    assert max(len(s.send) for s in states) == 7
    top = amg.ops[0].top
    assert top.matrix_info(0, "Apre")["lanes"] == 1 and top.matrix_info(0, "Q")["fmt"] == "sellwin"
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi", threads=_threads()).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)
