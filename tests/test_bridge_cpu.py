"""Bridge from the reference's element-partitioned layout (duplicated interface dofs, partial-sum rows, master = lowest
rank: dcc_map.cpp:497-543, hybrid_matrix.cpp:17-307) to the owner-row layout, on synthetic 2 x 2 (x 1) and 2 x 2 x 2 partitions."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from ngsamg_amd import bridge as B
from ngsamg_amd import dist as D
from oracle.pyoracle import Oracle
from tests.dist_cpu_backend import cpu_backend


def _setup(pgrid, gshape):
    R = int(np.prod(pgrid))
    comm = D.LoopbackComm(R)
    locs, gids = zip(*[B.shared_poisson_partition(r, pgrid, gshape) for r in range(R)])
    states, vmaps = B.from_shared_layout(comm, list(locs))
    return comm, locs, gids, states, vmaps


def _global_matrix(gshape):
    st = D.assemble_poisson_owned(0, (1,) * len(gshape), gshape)
    return sp.csr_matrix(st.A), st.free


@pytest.mark.parametrize("pgrid,gshape", [((2, 2, 1), (7, 8, 5)), ((2, 2, 2), (7, 7, 7)), ((3, 1, 2), (8, 4, 6)), ((2, 2), (11, 9))])
def test_owner_rows_equal_the_global_matrix(pgrid, gshape):
    comm, locs, gids, states, vmaps = _setup(pgrid, gshape)
    Ag, free_g = _global_matrix(gshape)
    n = Ag.shape[0]
    # every global vertex has exactly one owner, the lowest rank holding it
    owner_gid = [g[vm.perm[:vm.n_own]] for g, vm in zip(gids, vmaps)]
    allg = np.concatenate(owner_gid)
    assert np.array_equal(np.sort(allg), np.arange(n))
    for r, (L, g) in enumerate(zip(locs, gids)):
        m = np.array([min([r] + list(p)) for p in L.dist_procs])
        assert np.array_equal(np.sort(g[m == r]), np.sort(owner_gid[r]))
    # owned rows, columns mapped to global vertex ids, equal the rows of the serially assembled matrix
    for r, st in enumerate(states):
        cmap = np.concatenate([owner_gid[r], np.array([owner_gid[o][i] for o, i in zip(st.ghost_owner, st.ghost_rindex)], dtype=np.int64)])
        A = sp.csr_matrix(st.A).tocoo()
        Ar = sp.coo_matrix((A.data, (A.row, cmap[A.col])), shape=(st.n, n)).tocsr()
        ref = Ag[owner_gid[r]]
        assert abs(Ar - ref).max() < 1e-12 * abs(ref).max()
        assert np.array_equal(st.free, free_g[owner_gid[r]])
        assert np.all(np.diff(st.ghost_owner) >= 0) and r not in st.ghost_owner


@pytest.mark.parametrize("pgrid,gshape", [((2, 2, 1), (7, 8, 5)), ((2, 2, 2), (7, 7, 7))])
def test_distributed_vector_becomes_owner_vector(pgrid, gshape):
    """DISTRIBUTED (partial sums on the shared dofs) -> owner layout by one ghost -> owner add on the shared-dof map"""
    comm, locs, gids, states, vmaps = _setup(pgrid, gshape)
    n = int(np.prod(gshape))
    rng = np.random.default_rng(0)
    bg = rng.standard_normal(n)
    # split every global value randomly among the ranks that hold the vertex
    shares = [rng.uniform(0.2, 1.0, size=L.n_loc) for L in locs]
    tot = np.zeros(n)
    for g, s in zip(gids, shares):
        np.add.at(tot, g, s)
    loc_vecs = [bg[g] * s / tot[g] for g, s in zip(gids, shares)]
    owned = B.accumulate_host(comm, vmaps, [vm.to_ext(v) for vm, v in zip(vmaps, loc_vecs)])
    for g, vm, o in zip(gids, vmaps, owned):
        assert np.allclose(o, bg[g[vm.perm[:vm.n_own]]], rtol=1e-13, atol=1e-13)
    # round trip of the numbering
    for L, vm, v in zip(locs, vmaps, loc_vecs):
        assert np.array_equal(vm.from_ext(vm.to_ext(v), L.n_loc), v)


def test_bridged_partition_runs_the_distributed_cycle():
    """the converted states drive DistributedAMG like the natively assembled ones; result = serial oracle on the global hierarchy"""
    comm, locs, gids, states, vmaps = _setup((2, 2, 2), (13, 13, 13))
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=60, backend=cpu_backend(), max_coarse_size=10)
    rng = np.random.default_rng(1)
    bs = [torch.from_numpy(rng.standard_normal(s.n) * s.free) for s in states]
    xs = [torch.zeros(s.n, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    ref = Oracle(amg.global_levels(), sm_type="jacobi").apply(np.concatenate([b.numpy() for b in bs]))
    got = np.concatenate([x.numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)


def test_hybrid_smoother_weight_uses_partial_sums_on_the_bridged_level():
    """hybrid_smoother_utils.hpp:74-103: the off-rank weight ad_k adds |g_kj| of the LOCAL matrices of all ranks, i.e. the
    sum over the ranks of the absolute partial values -- not |assembled g_kj|.  The bridge keeps that sum for level 0."""
    from tests.dist_cpu_backend import cpu_backend as be
    comm, locs, gids, states, vmaps = _setup((2, 2, 1), (9, 9, 4))
    # ground truth from the reference's layout: for every owned dof k and every column owned elsewhere, sum_r |A_r(k, j)|
    gp = [s.G_abs_partial.copy() for s in states]
    tot_abs = sum(float(g.sum()) for g in gp)
    tot_asm = sum(float(abs(sp.csr_matrix(s.A)[:, s.n:]).sum()) for s in states)
    assert tot_abs >= tot_asm - 1e-12 and tot_abs > 0
    # independent count on the raw partial matrices: entries (k, j) of rank r with k not interior to r ...
    R = len(locs)
    master = [np.array([min([L.rank] + [int(q) for q in p]) for p in L.dist_procs]) for L in locs]
    raw = 0.0
    for L, m in zip(locs, master):
        A = sp.coo_matrix(L.A)
        raw += float(np.abs(A.data[m[A.row] != m[A.col]]).sum())          # couplings whose ends have different owners
    assert abs(raw - tot_abs) <= 1e-10 * raw
    # a partition whose partial sums cancel: flip the sign pattern of one rank's interface couplings artificially
    locs2 = [B.SharedLocal(L.rank, L.A.copy(), L.dist_procs, L.free, L.coords, getattr(L, "ex_key", None)) for L in locs]
    A0 = sp.lil_matrix(locs2[0].A)
    sh = np.array([len(p) > 0 for p in locs2[0].dist_procs])
    idx = np.flatnonzero(sh)
    for k in idx[:6]:
        for j in idx[:6]:
            if k != j and A0[k, j] != 0:
                A0[k, j] = -3.0 * A0[k, j]
    locs2[0].A = sp.csr_matrix(A0)
    st2, _ = B.from_shared_layout(comm, locs2)
    d_abs = sum(float(s.G_abs_partial.sum()) for s in st2)
    d_asm = sum(float(abs(sp.csr_matrix(s.A)[:, s.n:]).sum()) for s in st2)
    assert d_abs >= d_asm
    # and the distributed hybrid GS picks it up on level 0 (and only there)
    amg = D.DistributedAMG(comm, st2, dim=3, dist_min_rows=20, backend=be(sm_type="gs"), max_coarse_size=10, sm_type="gs")
    s0 = amg.dist_levels[0][0]
    assert getattr(s0, "G_abs_partial", None) is not None and s0.G_abs_partial.shape == (s0.n, s0.ghost_owner.size)
    assert getattr(amg.dist_levels[1][0], "G_abs_partial", None) is None


def _with_empty_master(pgrid, gshape):
    """the partition of _setup shifted to ranks 1..R with an EMPTY rank 0 in front (NGSolve's classic MPI layout: the master
    rank holds no part of the mesh; the reference guards its smoothers with `if (A.Height())`)"""
    R = int(np.prod(pgrid)) + 1
    comm = D.LoopbackComm(R)
    locs = []
    for r in range(R - 1):
        L, _ = B.shared_poisson_partition(r, pgrid, gshape)
        L.rank = r + 1
        L.dist_procs = [np.asarray(p) + 1 for p in L.dist_procs]
        locs.append(L)
    empty = B.SharedLocal(0, sp.csr_matrix((0, 0)), [], free=np.zeros(0, dtype=np.uint8), coords=np.zeros((0, len(gshape))))
    states, vmaps = B.from_shared_layout(comm, [empty] + locs)
    return comm, states, vmaps


@pytest.mark.parametrize("sm", ["jacobi", "gs", "hgs"])
def test_rank_without_rows_takes_part_in_the_collective_cycle(sm):
    from tests.dist_oracle import oracle_sm_types
    comm, states, vmaps = _with_empty_master((2, 2, 1), (13, 13, 6))
    assert states[0].n == 0 and all(s.n > 0 for s in states[1:])
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=30, backend=cpu_backend(sm_type=sm), max_coarse_size=10, sm_type=sm,
                           hgs_block_rows=32, gs_stage_min_rows=20)
    assert amg.k >= 1 and amg.dist_levels[1][0].n == 0
    rng = np.random.default_rng(1)
    bs = [torch.from_numpy(rng.standard_normal(s.n) * s.free) for s in states]
    xs = [torch.zeros(s.n, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg)).apply(np.concatenate([b.numpy() for b in bs]))
    got = np.concatenate([x.numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


def test_energy_flag_of_the_reference():
    """ngs_amg_energy (amg_pc.cpp:333): alg on an elasticity preconditioner = the reference's edge-matrix setup, elmat is refused"""
    from ngsamg_amd.NgsAMG import _energy_flag, _flags
    from ngsamg_amd._lib import NgsAMGError
    assert _energy_flag(_flags({}), 1) == {}
    assert _energy_flag(_flags({"ngs_amg_energy": "alg"}), 0) == {}
    assert _energy_flag(_flags({"ngs_amg_energy": "alg"}), 1) == {"edge_mats": 1}
    assert _energy_flag(_flags({"ngs_amg_energy": "alg", "ngs_amg_edge_mats": False}), 1) == {}
    assert _energy_flag(_flags({"ngs_amg_energy": "triv"}), 1) == {}
    for bad in ("elmat", "nonsense"):
        try:
            _energy_flag(_flags({"ngs_amg_energy": bad}), 1)
            assert False
        except NgsAMGError:
            pass
