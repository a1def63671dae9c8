"""Rank-partitioned V-cycle on CPU: loopback (virtual ranks in one process) and world_size-2 gloo processes.
The distributed result must equal the serial oracle on the assembled global hierarchy."""
import os

import numpy as np
import pytest
import torch

from ngsamg_amd import dist as D
from oracle.pyoracle import Oracle
from tests.dist_cpu_backend import cpu_backend
from tests.dist_oracle import oracle_bgs, oracle_sm_types


def _run_loopback(R, box, dim, dist_min_rows, sm="jacobi", fold=True, **extra):
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dist_min_rows, backend=cpu_backend(sm_type=sm),
                           max_coarse_size=10, sm_type=sm, fold=fold, **extra)
    rng = np.random.default_rng(0)
    bs = [torch.from_numpy(rng.standard_normal(s.n) * s.free) for s in states]
    xs = [torch.zeros(s.n, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv)).apply(np.concatenate([b.numpy() for b in bs]))
    got = np.concatenate([x.numpy() for x in xs])
    return amg, got, ref


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (12, 12, 12), 3, 100), (4, (10, 10, 10), 3, 50), (8, (8, 8, 8), 3, 20),
                                            (4, (24, 24), 2, 50), (2, (12, 12, 12), 3, 10 ** 9),
                                            (5, (6, 8, 8), 3, 50), (6, (7, 6, 8), 3, 50)])   # the reference runs NP = 2 and 5
@pytest.mark.parametrize("fold", [True, False])
def test_loopback_matches_serial_oracle(R, box, dim, dmin, fold):
    """fold = True: stages as the single-GPU cycle runs them (z = S(S0 b) on the way down, x = z + Q x_c on the way up);
    fold = False: the literal stage sequence.  Both must equal the serial oracle's literal cycle."""
    amg, got, ref = _run_loopback(R, box, dim, dmin, fold=fold)
    assert amg.k >= 1 and amg.fold == fold
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (12, 12, 12), 3, 100), (4, (10, 10, 10), 3, 50), (8, (8, 8, 8), 3, 20), (4, (24, 24), 2, 50)])
def test_loopback_hybrid_gs_matches_serial_hybrid_oracle(R, box, dim, dmin):
    """rank-partitioned hybrid Gauss-Seidel == the oracle's serial hybrid GS (blocks = ranks, modified diagonal)"""
    amg, got, ref = _run_loopback(R, box, dim, dmin, "gs")
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
    # and the hybrid smoother still gives a convergent preconditioner: PCG on the global system
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg))
    rng = np.random.default_rng(5)
    b = rng.standard_normal(glv[0].A.n_rows) * glv[0].free
    _, it, errs = orc.pcg(b, tol=1e-8, maxit=80)
    assert errs[-1] < 1e-8 * errs[0] and it < 60


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (12, 12, 12), 3, 100), (4, (10, 10, 10), 3, 50), (8, (8, 8, 8), 3, 20)])
def test_loopback_hybrid_block_gs_matches_serial_hybrid_oracle(R, box, dim, dmin):
    """rank-partitioned hybrid BLOCK Gauss-Seidel (reference HybridBS: BSmoother on the local matrix with the l1-modified
    diagonal, blocks = local aggregates) == the oracle's serial block smoother with frozen off-rank values"""
    amg, got, ref = _run_loopback(R, box, dim, dmin, "bgs")
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv))
    rng = np.random.default_rng(5)
    b = rng.standard_normal(glv[0].A.n_rows) * glv[0].free
    _, it, errs = orc.pcg(b, tol=1e-8, maxit=80)
    assert errs[-1] < 1e-8 * errs[0] and it < 60


def test_partitioned_matrix_is_the_global_one():
    """owned rows of all ranks, renumbered globally, give a symmetric matrix with zero row sums"""
    R, box = 4, (9, 9, 9)
    pg = D.proc_grid(R, 3)
    comm = D.LoopbackComm(R)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=50, backend=cpu_backend(), max_coarse_size=10)
    A = amg.global_levels()[0].A.to_scipy()
    assert abs(A - A.T).max() < 1e-13
    assert np.abs(A.sum(axis=1)).max() < 1e-12
    for l, lv in enumerate(amg.dist_levels):
        assert sum(s.n for s in lv) > 0
        for s in lv:
            assert np.all(np.diff(s.ghost_owner) >= 0)          # ghosts sorted by owner
            assert s.rank not in s.ghost_owner                  # no self ghosts


def _worker(rank, world, port, box, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = D.TorchComm()
        pg = D.proc_grid(world, 3)
        st = D.assemble_poisson_owned(rank, pg, box)
        amg = D.DistributedAMG(comm, [st], dim=3, dist_min_rows=100, backend=cpu_backend(), max_coarse_size=10)
        rng = np.random.default_rng(rank)
        b = torch.from_numpy(rng.standard_normal(st.n) * st.free)
        x = torch.zeros(st.n, dtype=torch.float64)
        amg.Mult([b], [x])
        glv = amg.global_levels()
        allb = [None] * world
        allx = [None] * world
        dist.all_gather_object(allb, b.numpy())
        dist.all_gather_object(allx, x.numpy())
        if rank == 0:
            ref = Oracle(glv, sm_type="jacobi").apply(np.concatenate(allb))
            got = np.concatenate(allx)
            q.put(float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    finally:
        dist.destroy_process_group()


def test_two_gloo_processes_match_serial_oracle():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, (12, 12, 12), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    for p in procs:
        assert p.exitcode == 0
    assert q.get(timeout=10) < 1e-12


# ---- committed fixtures of the hybrid smoothers on synthetic partitions (tests/golden/make_golden_hybrid.py) ---------------

HYBRID_CASES = {"hybrid_poisson2d_2x2": dict(R=4, box=(9, 9), dim=2, dmin=30),
                "hybrid_poisson3d_2x2x2": dict(R=8, box=(5, 5, 5), dim=3, dmin=20)}


def _hybrid_fixture(name):
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


@pytest.mark.parametrize("name", sorted(HYBRID_CASES))
@pytest.mark.parametrize("sm,tol", [("jacobi", 1e-12), ("gs", 1e-10), ("bgs", 1e-10)])
def test_distributed_setup_and_cycle_reproduce_hybrid_fixture(name, sm, tol):
    """partition, distributed coarsening, halo tables and the stage sequence are pinned: a fresh loopback run gives the
    fixture's V-cycle vector"""
    z = _hybrid_fixture(name)
    c = HYBRID_CASES[name]
    comm = D.LoopbackComm(c["R"])
    pg = D.proc_grid(c["R"], c["dim"])
    states = [D.assemble_poisson_owned(r, pg, c["box"]) for r in range(c["R"])]
    assert np.array_equal([s.n for s in states], z["rank_sizes"])
    amg = D.DistributedAMG(comm, states, dim=c["dim"], dist_min_rows=c["dmin"], backend=cpu_backend(sm_type=sm),
                           max_coarse_size=10, sm_type=sm)
    assert amg.k == int(z[f"{sm}_k"])
    off = np.concatenate([[0], np.cumsum(z["rank_sizes"])])
    bs = [torch.from_numpy(z["b"][off[r]:off[r + 1]].copy()) for r in range(c["R"])]
    xs = [torch.zeros(s.n, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    got = np.concatenate([x.numpy() for x in xs])
    ref = z[f"{sm}_V"]
    assert np.linalg.norm(got - ref) <= tol * np.linalg.norm(ref)


@pytest.mark.parametrize("name", sorted(HYBRID_CASES))
def test_serial_hybrid_gs_oracle_reproduces_fixture_from_stored_arrays(name):
    """the oracle's serial hybrid Gauss-Seidel on the STORED global hierarchy (matrices, modified diagonals, rank of
    every row, visiting order) -- no setup code involved"""
    from types import SimpleNamespace
    from ngsamg_amd._lib import Matrix
    z = _hybrid_fixture(name)
    nl, k = int(z["n_levels"]), int(z["gs_k"])

    def mat(l, tag):
        key = f"l{l}_{tag}_shape"
        if key not in z:
            return None
        nr, nc, br, bc = (int(v) for v in z[key])
        return Matrix(nr, nc, br, bc, z[f"l{l}_{tag}_rowptr"], z[f"l{l}_{tag}_col"], z[f"l{l}_{tag}_val"])

    levels = []
    for l in range(nl):
        L = SimpleNamespace(A=mat(l, "A"), P=mat(l, "P"), PT=mat(l, "PT"), free=z[f"l{l}_free"], dinv=z[f"l{l}_dinv"],
                            color=z[f"l{l}_color"], coords=None, agg=None)
        if l < k:
            L.dinv, L.gs_order, L.gs_block = z[f"gs_l{l}_dinv"], z[f"gs_l{l}_gs_order"], z[f"gs_l{l}_gs_block"]
        levels.append(L)
    # (the replicated tail levels carry the colours of the host setup: 'gs_mc' uses them)
    got = Oracle(levels, sm_type=["gs_order"] * k + ["gs_mc"] * (nl - k)).apply(z["b"])
    assert np.linalg.norm(got - z["gs_V"]) <= 1e-12 * np.linalg.norm(z["gs_V"])


def test_allgather_compaction_index():
    """the RCCL all-gather works on pieces padded to the longest one; one gather with this index removes the padding"""
    counts = [5, 3, 0, 4]
    m = max(counts)
    pieces = [np.arange(c) + 10 * r for r, c in enumerate(counts)]
    buf = np.full(m * len(counts), -1.0)
    for r, pc in enumerate(pieces):
        buf[r * m:r * m + pc.size] = pc
    assert np.array_equal(buf[D.compaction_index(counts, m)], np.concatenate(pieces))


@pytest.mark.parametrize("R,box,rot", [(2, (6, 5, 5), False), (4, (5, 5, 4), False), (2, (5, 5, 4), True), (8, (4, 4, 4), False)])
def test_loopback_elasticity_matches_serial_oracle(R, box, rot):
    """rank-partitioned ELASTICITY levels (3x3 fine blocks -> 6x6 coarse blocks with rigid-body prolongation blocks, or 6x6
    everywhere with rotations): block-Jacobi in the literal stage order == the serial oracle on the assembled global hierarchy"""
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_elasticity_owned(r, pg, box, rotations=rot) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=10, backend=cpu_backend(), max_coarse_size=5, energy=1,
                           regularize_cmats=0 if rot else 1)
    assert amg.k >= 1 and not amg.fold
    bs0 = states[0].bs
    assert bs0 == (6 if rot else 3) and amg.dist_levels[1][0].bs == 6
    rng = np.random.default_rng(0)
    bs = [torch.from_numpy(rng.standard_normal(s.n * bs0) * np.repeat(s.free, bs0)) for s in states]
    xs = [torch.zeros(s.n * bs0, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    glv = amg.global_levels()
    assert glv[0].A.br == bs0 and glv[0].P.bc == 6
    ref = Oracle(glv, sm_type="jacobi").apply(np.concatenate([b.numpy() for b in bs]))
    got = np.concatenate([x.numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-11 * np.linalg.norm(ref)
    # the assembled level-0 matrix is symmetric and annihilates translations
    A = glv[0].A.to_scipy()
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()
    t = np.zeros((A.shape[0] // bs0, bs0))
    t[:, 0] = 1.0
    assert np.abs(A @ t.ravel()).max() < 1e-11


@pytest.mark.parametrize("R,box,rot", [(2, (6, 5, 5), False), (4, (5, 5, 4), True), (8, (4, 4, 4), False)])
def test_loopback_elasticity_hybrid_gs_matches_serial_hybrid_oracle(R, box, rot):
    """rank-partitioned elasticity levels with the HYBRID (block-)Gauss-Seidel smoother: every rank sweeps its block rows,
    off-rank couplings frozen, block diagonal modified by max_l 0.51 (1 + ad(l)) (hybrid_smoother_utils.hpp:86-141) -- the
    convergent smoother for the rotational problem (cfg 5), where block-Jacobi stagnates"""
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_elasticity_owned(r, pg, box, rotations=rot) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=10, backend=cpu_backend(sm_type="gs"), max_coarse_size=5, energy=1,
                           regularize_cmats=0 if rot else 1, sm_type="gs", gs_stage_min_rows=20)
    assert amg.k >= 1
    bs0 = states[0].bs
    # the modified diagonal differs from the plain one exactly where off-rank couplings are strong
    s0 = amg.dist_levels[0][0]
    d_plain = s0.dinv_ext[:s0.n * bs0 * bs0].reshape(s0.n, -1)
    d_mod = s0.dinv_gs_ext[:s0.n * bs0 * bs0].reshape(s0.n, -1)
    ratio = np.abs(d_plain).sum(axis=1) / np.maximum(np.abs(d_mod).sum(axis=1), 1e-300)
    fr = s0.free.astype(bool)
    assert np.all(ratio[fr] >= 1.0 - 1e-12) and ratio[fr].max() > 1.0 + 1e-6
    assert np.allclose(ratio[:s0.n_interior][fr[:s0.n_interior]], 1.0)       # interior rows have no off-rank couplings
    rng = np.random.default_rng(0)
    bs = [torch.from_numpy(rng.standard_normal(s.n * bs0) * np.repeat(s.free, bs0)) for s in states]
    xs = [torch.zeros(s.n * bs0, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg)).apply(np.concatenate([b.numpy() for b in bs]))
    got = np.concatenate([x.numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-11 * np.linalg.norm(ref)
    # and it is a convergent preconditioner: PCG on the assembled operator
    A = glv[0].A
    bg = np.concatenate([b.numpy() for b in bs])
    its = Oracle(glv, sm_type=oracle_sm_types(amg)).pcg(bg, tol=1e-8, maxit=60)[1]
    assert its < 40


@pytest.mark.parametrize("R,box,dim,dmin,B", [(2, (12, 12, 12), 3, 100, 64), (4, (10, 10, 10), 3, 50, 32), (8, (8, 8, 8), 3, 20, 64), (4, (24, 24), 2, 50, 128),
                                              (2, (30, 30, 30), 3, 10000, 256)])          # last: block-hybrid levels in the replicated tail too
def test_loopback_block_hybrid_gs_matches_serial_hybrid_oracle(R, box, dim, dmin, B):
    """sm_type = hgs: block-hybrid Gauss-Seidel on rank-partitioned levels (blocks of B consecutive owned rows, l1-modified
    diagonal incl. the couplings to ghost columns) == the oracle's serial hybrid GS with the same blocks and order"""
    amg, got, ref = _run_loopback(R, box, dim, dmin, "hgs", hgs_block_rows=B)
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg))
    rng = np.random.default_rng(5)
    b = rng.standard_normal(glv[0].A.n_rows) * glv[0].free
    _, it, errs = orc.pcg(b, tol=1e-8, maxit=80)
    assert errs[-1] < 1e-8 * errs[0] and it < 60


# ---- strong scaling: ONE global grid cut into balanced, unequal pieces (bench.py --gpus N) ------------------------------

def _strong_states(pg, gshape, coords="rng"):
    R = int(np.prod(pg))
    return [D.assemble_poisson_owned(r, pg, None, gshape=gshape, coords=coords) for r in range(R)]


@pytest.mark.parametrize("pg,gshape", [((3, 1, 1), (11, 9, 10)), ((2, 3, 2), (11, 9, 10)), ((8, 1, 1), (27, 6, 5)), ((3, 2), (14, 11))])
def test_strong_split_is_the_single_gpu_matrix(pg, gshape):
    """the pieces of the balanced split, glued together, are the matrix fem.poisson_fast assembles for the whole grid
    (same vertex positions, same assembly): N > 1 of bench.py solves the problem N = 1 solves"""
    import scipy.sparse as sp
    from ngsamg_amd import fem
    p = fem.poisson_fast(gshape, dirichlet="right|top", jitter=0.2, seed=1)
    Ag = sp.csr_matrix((p.val, p.col, p.rowptr), shape=(p.n, p.n))
    dim = len(gshape)
    cuts = D.grid_cuts(pg, gshape)
    assert all(c[0] == 0 and c[-1] == g and np.all(np.diff(c) >= g // q) for c, g, q in zip(cuts, gshape, pg))
    sts = _strong_states(pg, gshape)
    assert sum(s.n for s in sts) == p.n
    gids = []
    for s in sts:
        pc = np.unravel_index(s.rank, pg)
        I = np.meshgrid(*[np.arange(cuts[d][pc[d]], cuts[d][pc[d] + 1]) for d in range(dim)], indexing="ij")
        gids.append(np.ravel_multi_index([i.reshape(-1) for i in I], gshape))
    for s in sts:
        gcol = np.concatenate([gids[s.rank], np.asarray([gids[o][ri] for o, ri in zip(s.ghost_owner, s.ghost_rindex)], dtype=np.int64)])
        A = sp.csr_matrix(s.A).tocoo()
        Ar = sp.csr_matrix((A.data, (A.row, gcol[A.col])), shape=(s.n, p.n))
        assert abs(Ar - Ag[gids[s.rank]]).max() < 1e-14
        assert np.array_equal(s.free, p.free[gids[s.rank]])


@pytest.mark.parametrize("pg,gshape,dmin", [((3, 1, 1), (20, 12, 12), 100), ((5, 1, 1), (23, 9, 10), 50), ((2, 2, 2), (15, 13, 11), 20)])
@pytest.mark.parametrize("sm", ["jacobi", "gs"])
def test_strong_split_cycle_matches_serial_oracle(pg, gshape, dmin, sm):
    """unequal pieces through the whole distributed setup and cycle (CPU stage backend) == serial oracle"""
    R = int(np.prod(pg))
    comm = D.LoopbackComm(R)
    states = _strong_states(pg, gshape)
    assert len({s.n for s in states}) > 1          # the pieces really differ in size
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=dmin, backend=cpu_backend(sm_type=sm), max_coarse_size=10, sm_type=sm)
    rng = np.random.default_rng(1)
    bs = [torch.from_numpy(rng.standard_normal(s.n) * s.free) for s in states]
    xs = [torch.zeros(s.n, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg)).apply(np.concatenate([b.numpy() for b in bs]))
    got = np.concatenate([x.numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= (1e-12 if sm == "jacobi" else 1e-10) * np.linalg.norm(ref)


@pytest.mark.parametrize("R,box,rot", [(2, (8, 8, 8), False), (2, (6, 6, 6), True), (4, (6, 6, 5), False)])
def test_loopback_elasticity_edge_matrix_prolongation(R, box, rot):
    """ngs_amg_edge_mats on rank-partitioned elasticity levels: the rank-local setup builds the matrix-valued smoothed prolongation
    (general 6x6 / 3x6 blocks) from the edge matrices of its owned block -- every rank-partitioned level derives them from its level
    matrix as the finest level does, the replicated tail carries them on -- and the halo rows of P travel as blocks.  Result ==
    serial oracle on the assembled hierarchy; that hierarchy is a convergent preconditioner."""
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_elasticity_owned(r, pg, box, rotations=rot) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=10, backend=cpu_backend(), max_coarse_size=5, energy=1,
                           regularize_cmats=0 if rot else 1, edge_mats=1)
    assert amg.k >= 1
    bs0 = states[0].bs
    rng = np.random.default_rng(0)
    bs = [torch.from_numpy(rng.standard_normal(s.n * bs0) * np.repeat(s.free, bs0)) for s in states]
    xs = [torch.zeros(s.n * bs0, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    glv = amg.global_levels()
    b = np.concatenate([v.numpy() for v in bs])
    ref = Oracle(glv, sm_type="jacobi").apply(b)
    got = np.concatenate([x.numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-11 * np.linalg.norm(ref)
    blk = np.asarray(glv[0].P.val).reshape(-1, glv[0].P.br, glv[0].P.bc)[:, :3, :3]
    assert np.abs(blk - np.eye(3) * blk[:, :1, :1]).max() > 1e-6          # general blocks, not w Q(t)
    _, it, errs = Oracle(glv, sm_type="gs").pcg(b, tol=1e-8, maxit=100)
    assert errs[-1] < 1e-8 * errs[0] and it < 40
