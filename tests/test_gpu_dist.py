"""Rank-partitioned V-cycle through the HIP library: several virtual ranks on ONE GPU (LoopbackComm), checked against
the serial oracle on the assembled global hierarchy.  The same DistributedAMG code runs one rank per process over
torch.distributed (RCCL) in bench.py --gpus N; only the transport differs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (4, (12, 12, 12), 3, 100), (8, (10, 10, 10), 3, 40),
                                            (4, (40, 40), 2, 100)])
@pytest.mark.parametrize("fold", [True, False])
def test_loopback_device_matches_serial_oracle(R, box, dim, dmin, fold):
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10, fold=fold)
    assert amg.k >= 1
    assert all(op.top.is_folded(l) == fold for op in amg.ops for l in range(amg.k))
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):          # second application re-uses all buffers
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi").apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)


def test_distributed_equals_single_gpu_cycle_on_global_hierarchy():
    """the assembled global hierarchy run through the ordinary single-GPU path gives the same vector"""
    import torch
    from ngsamg_amd import dist as D
    from ngsamg_amd.device import DeviceAMGMatrix
    R, box = 4, (14, 14, 14)
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=100, device=0, max_coarse_size=10)
    rng = np.random.default_rng(1)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
    amg.Mult(bs, xs)
    torch.cuda.synchronize()

    class H:
        pass
    h = H()
    h.levels = amg.global_levels()
    h.coarse_n, h.coarse_inv, h.n_levels = amg.tail_hier.coarse_n, amg.tail_hier.coarse_inv, len(h.levels)
    ser = DeviceAMGMatrix(h, sm_type="jacobi", device=0)
    x = np.empty(sum(s.n for s in states))
    ser.Mult(np.concatenate(bh), x)
    got = np.concatenate([t.cpu().numpy() for t in xs])
    assert np.linalg.norm(got - x) <= 1e-12 * np.linalg.norm(x)


@pytest.mark.parametrize("sm", ["gs", "bgs"])
@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (8, (10, 10, 10), 3, 40), (4, (40, 40), 2, 100)])
def test_loopback_device_hybrid_gs_matches_serial_hybrid_oracle(R, box, dim, dmin, sm):
    """hybrid Gauss-Seidel on rank-partitioned levels (a5): local multicolour sweeps with frozen off-rank values and
    the l1-type modified diagonal == the oracle's serial hybrid GS with blocks = ranks; sm = "bgs": the hybrid BLOCK
    smoother (reference HybridBS: blocks = local aggregates, modified diagonal inside the diagonal blocks)"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm)
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv)).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("name,R,box,dim,dmin", [("hybrid_poisson2d_2x2", 4, (9, 9), 2, 30), ("hybrid_poisson3d_2x2x2", 8, (5, 5, 5), 3, 20)])
@pytest.mark.parametrize("sm,tol", [("jacobi", 1e-12), ("gs", 1e-10), ("bgs", 1e-10)])
def test_loopback_device_reproduces_hybrid_fixture(name, R, box, dim, dmin, sm, tol):
    """the rank-partitioned GPU path against the committed fixtures of the synthetic 2x2 / 2x2x2 partitions"""
    import os
    import torch
    from ngsamg_amd import dist as D
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm)
    off = np.concatenate([[0], np.cumsum(z["rank_sizes"])])
    bs = [torch.from_numpy(z["b"][off[r]:off[r + 1]].copy()).cuda() for r in range(R)]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
    amg.Mult(bs, xs)
    torch.cuda.synchronize()
    got = np.concatenate([x.cpu().numpy() for x in xs])
    ref = z[f"{sm}_V"]
    assert np.linalg.norm(got - ref) <= tol * np.linalg.norm(ref)


def _run_check(*argv, timeout=600, extra_env=None):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the launcher's own watchdog (kills exactly its rank processes) fires before this timeout would kill only the launcher
    env = dict(os.environ, NGSAMG_CHECK_TIMEOUT=str(max(30, timeout - 90)), **(extra_env or {}))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_rccl_check.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env)
    assert r.returncode == 0 and "RCCL CHECK PASSED" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def test_rccl_world_size_one_elasticity():
    """the block path (6x6 coarse blocks) through a real RCCL communicator at world size 1"""
    _run_check("--world", "1", "--box", "10", "--elast", "3", "--dmin", "50")


@pytest.mark.parametrize("sm", ["jacobi", "gs"])
def test_rccl_world_size_one(sm):
    """The native driver over a real RCCL communicator on ONE GPU (child process): ncclCommInitRank, the all-gather of
    level k, and -- through a halo map whose only peer is the rank itself -- pack kernel + ncclSend / ncclRecv in one group
    straight into the ghost segment, and the add direction.  More ranks need more GPUs: tests/test_gpu_multi.py."""
    out = _run_check("--world", "1", "--box", "20", "--sm", sm)
    assert "self-loop halo ok = True" in out
    # the whole collective cycle (RCCL operations included) was captured into a hipGraph and replayed
    assert "graph enabled = True graphs = 1 replays = 2" in out, out        # (the first application is launched directly)


@pytest.mark.parametrize("mode", ["1", "pad"])
@pytest.mark.parametrize("graph", ["1", "0"])
def test_rccl_world_size_one_runs_the_allgather_branch(mode, graph):
    """CtrMap's role (dof_contract.cpp:68-223): level k is gathered by ncclAllGather.  World size 1 normally takes a copy
    shortcut; AMGX_DIST_FORCE_ALLGATHER sends it through ncclAllGather ("1") and through the padded all-gather + compaction
    kernel of unequal pieces ("pad"), captured in the cycle's graph and with direct launches."""
    out = _run_check("--world", "1", "--box", "20", "--sm", "jacobi", extra_env={"AMGX_DIST_FORCE_ALLGATHER": mode, "AMGX_DIST_GRAPH": graph})
    assert f"allgather = {mode}" in out
    assert ("graph enabled = True graphs = 1 replays = 2" in out) == (graph == "1"), out


def _halo_tables(R, n, rng):
    """random symmetric exchange pattern between R virtual ranks: rank p ghosts a random subset of every other rank's rows"""
    want = {(p, q): np.sort(rng.choice(n[q], size=int(rng.integers(0, max(1, n[q] // 3))), replace=False)).astype(np.int32)
            for p in range(R) for q in range(R) if p != q}
    return want


@pytest.mark.parametrize("bs", [1, 3])
def test_halo_exchange_local_ranks(bs):
    """amgx_halo_exchange = DCCMap: owner -> ghost overwrite (CO2CU) and ghost -> owner add with zeroed ghosts (DIS2CO),
    hand-written pack / unpack kernels, R virtual ranks on one GPU; checked against numpy"""
    import ctypes as C
    import torch
    from ngsamg_amd import _lib
    lib = _lib.hip()
    rng = np.random.default_rng(7)
    R = 4
    n = [int(v) for v in rng.integers(50, 400, size=R)]
    want = _halo_tables(R, n, rng)
    comm = C.c_void_p()
    assert lib.amgx_comm_create(_lib.AMGX_COMM_LOCAL, R, 0, None, 0, C.byref(comm)) == 0
    halos, keep, vecs, host = [], [], [], []
    for p in range(R):
        peers = np.array([q for q in range(R) if q != p], dtype=np.int32)
        send = [want[(q, p)] for q in peers]            # rows of mine that q ghosts
        recv = [want[(p, q)].size for q in peers]
        sp = np.concatenate([[0], np.cumsum([v.size for v in send])]).astype(np.int64)
        rp = np.concatenate([[0], np.cumsum(recv)]).astype(np.int64)
        si = np.ascontiguousarray(np.concatenate(send), dtype=np.int32)
        d = _lib.amgx_halo_desc()
        d.n_peers = peers.size
        d.peer_rank, d.send_ptr, d.send_idx, d.recv_ptr = _lib.ptr(peers, C.c_int32), _lib.ptr(sp, C.c_int64), _lib.ptr(si, C.c_int32), _lib.ptr(rp, C.c_int64)
        h = C.c_void_p()
        assert lib.amgx_halo_create(comm, C.byref(d), n[p], int(rp[-1]), bs, p, C.byref(h)) == 0, lib.amgx_comm_last_error(comm)
        halos.append(h)
        keep += [peers, sp, rp, si]
        v = rng.standard_normal((n[p] + int(rp[-1])) * bs)
        host.append(v.reshape(-1, bs).copy())
        vecs.append(torch.from_numpy(v).cuda())
    hp = (C.c_void_p * R)(*halos)
    vp = (C.c_void_p * R)(*[v.data_ptr() for v in vecs])
    assert lib.amgx_halo_exchange(comm, R, hp, vp, 0) == 0, lib.amgx_comm_last_error(comm)
    lib.amgx_comm_synchronize(comm)
    exp = [h.copy() for h in host]
    for p in range(R):
        off = n[p]
        for q in range(R):
            if q == p:
                continue
            w = want[(p, q)]
            exp[p][off:off + w.size] = host[q][w]
            off += w.size
    for p in range(R):
        assert np.array_equal(vecs[p].cpu().numpy().reshape(-1, bs), exp[p])
    assert lib.amgx_halo_exchange(comm, R, hp, vp, 1) == 0, lib.amgx_comm_last_error(comm)
    lib.amgx_comm_synchronize(comm)
    exp2 = [h.copy() for h in exp]
    for p in range(R):
        off = n[p]
        for q in range(R):
            if q == p:
                continue
            w = want[(p, q)]
            np.add.at(exp2[q], w, exp[p][off:off + w.size])
            off += w.size
        exp2[p][n[p]:] = 0.0
    for p in range(R):
        assert np.allclose(vecs[p].cpu().numpy().reshape(-1, bs), exp2[p], rtol=1e-15, atol=0)
    for h in halos:
        lib.amgx_halo_destroy(h)
    lib.amgx_comm_destroy(comm)


def test_overlap_split_equals_unsplit(monkeypatch):
    """interior rows processed while the exchange is in flight + boundary rows afterwards == one pass over all rows; also
    b_status = 0 (DISTRIBUTED right-hand side: ghost entries are contributions to the owners, added first)"""
    import torch
    from ngsamg_amd import dist as D
    R, box = 4, (14, 14, 14)
    res = []
    for no_overlap in (False, True):
        if no_overlap:
            monkeypatch.setenv("AMGX_DIST_NO_OVERLAP", "1")
        comm = D.LoopbackComm(R)
        states = [D.assemble_poisson_owned(r, D.proc_grid(R, 3), box) for r in range(R)]
        amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=100, device=0, max_coarse_size=10)
        assert all(s.n_interior > 0 for s in states)
        rng = np.random.default_rng(2)
        bh = [rng.standard_normal(s.n) * s.free for s in states]
        xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
        amg.Mult([torch.from_numpy(b).cuda() for b in bh], xs)
        torch.cuda.synchronize()
        res.append(np.concatenate([x.cpu().numpy() for x in xs]))
        if not no_overlap:
            # the same right-hand side handed over in DISTRIBUTED form: every owner keeps a random share of its entries that
            # a peer ghosts, the rest sits in the peers' ghost entries
            ext = [np.concatenate([b, np.zeros(s.ghost_owner.size)]) for b, s in zip(bh, states)]
            for p, s in enumerate(states):
                for q, (a, e) in s.recv_seg.items():
                    rows = states[q].send[p]
                    share = rng.uniform(0.1, 0.9, size=rows.size)
                    ext[p][s.n + a:s.n + e] = share * bh[q][rows]
                    ext[q][rows] -= share * bh[q][rows]
            xd = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
            amg.Mult([torch.from_numpy(v).cuda() for v in ext], xd, b_status=0)
            torch.cuda.synchronize()
            got = np.concatenate([x.cpu().numpy() for x in xd])
            assert np.linalg.norm(got - res[0]) <= 1e-12 * np.linalg.norm(res[0])
    assert np.linalg.norm(res[0] - res[1]) <= 1e-14 * np.linalg.norm(res[1])


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (4, (12, 12, 12), 3, 100), (3, (20, 9, 9), 3, 100), (1, (20, 18, 16), 3, 200)])
def test_loopback_device_fused_kernels_on_rank_partitioned_levels(R, box, dim, dmin, monkeypatch):
    """the production shape of the multi-GPU run: level 0 in the one-thread-per-row form, i.e. the FUSED down kernel
    (pre-smoothing + chunk-local restriction, omega*Dinv in the diagonal slot) and the windowed Q kernel on matrices WITH
    ghost columns -- forced onto small levels"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_SELL_MAX_LANES", "1")
    # (R = 1: a rank without ghost columns has SQUARE levels -- the compact-chunk form of the fused kernel must stay off on a handle
    #  that is driven in interior / boundary chunk ranges; the size threshold is lowered so that it would otherwise apply)
    monkeypatch.setenv("AMGX_COMPACT_CHUNKS_MIN_ROWS", "100")
    comm = D.LoopbackComm(R)
    pg = (R, 1, 1) if R == 3 else D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10)
    top = amg.ops[0].top
    assert top.matrix_info(0, "Apre")["lanes"] == 1 and top.matrix_info(0, "Q")["fmt"] is not None
    assert top.time_op(0, 7, reps=2) > 0            # the fused kernel exists on the rank-partitioned level 0
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi").apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)


def test_bridged_reference_layout_on_device():
    """NGSolve-style partition (duplicated interface dofs, partial-sum rows) -> owner rows (ngsamg_amd/bridge.py) -> native
    distributed cycle; the DISTRIBUTED right-hand side is brought to the owners by amgx_halo_exchange(mode 1) on the
    shared-dof map (DCCMap DIS2CO), the CUMULATED solution is returned to all copies by mode 0 (CO2CU)."""
    import ctypes as C
    import torch
    from ngsamg_amd import _lib, bridge as B, dist as D
    from oracle.pyoracle import Oracle
    pgrid, gshape = (2, 2, 2), (15, 15, 15)
    R = 8
    comm = D.LoopbackComm(R)
    locs, gids = zip(*[B.shared_poisson_partition(r, pgrid, gshape) for r in range(R)])
    states, vmaps = B.from_shared_layout(comm, list(locs))
    n = int(np.prod(gshape))
    rng = np.random.default_rng(4)
    free_g = np.ones(n)
    for L, g in zip(locs, gids):
        free_g[g] = L.free
    bg = rng.standard_normal(n) * free_g
    shares = [rng.uniform(0.2, 1.0, size=L.n_loc) for L in locs]
    tot = np.zeros(n)
    for g, s in zip(gids, shares):
        np.add.at(tot, g, s)
    loc_b = [bg[g] * s / tot[g] for g, s in zip(gids, shares)]          # DISTRIBUTED local vectors
    # ---- shared-dof map on the device
    lib = _lib.hip()
    cc = C.c_void_p()
    assert lib.amgx_comm_create(_lib.AMGX_COMM_LOCAL, R, 0, None, 0, C.byref(cc)) == 0
    halos, keep, vecs = [], [], []
    for r, vm in enumerate(vmaps):
        d = _lib.amgx_halo_desc()
        d.n_peers = vm.peers.size
        d.peer_rank, d.send_ptr, d.send_idx, d.recv_ptr = _lib.ptr(vm.peers, C.c_int32), _lib.ptr(vm.send_ptr, C.c_int64), _lib.ptr(vm.send_idx, C.c_int32), _lib.ptr(vm.recv_ptr, C.c_int64)
        h = C.c_void_p()
        assert lib.amgx_halo_create(cc, C.byref(d), vm.n_own, vm.n_ext - vm.n_own, 1, r, C.byref(h)) == 0, lib.amgx_comm_last_error(cc)
        halos.append(h)
        vecs.append(torch.from_numpy(vm.to_ext(loc_b[r])).cuda())
    hp = (C.c_void_p * R)(*halos)
    vp = (C.c_void_p * R)(*[v.data_ptr() for v in vecs])
    assert lib.amgx_halo_exchange(cc, R, hp, vp, 1) == 0, lib.amgx_comm_last_error(cc)
    lib.amgx_comm_synchronize(cc)
    ref_owned = B.accumulate_host(comm, vmaps, [vm.to_ext(v) for vm, v in zip(vmaps, loc_b)])
    for r, vm in enumerate(vmaps):
        assert np.allclose(vecs[r][:vm.n_own].cpu().numpy(), ref_owned[r], rtol=1e-14, atol=1e-14)
        assert np.allclose(ref_owned[r], bg[gids[r][vm.perm[:vm.n_own]]], rtol=1e-13, atol=1e-13)
    # ---- the cycle on the converted hierarchy (interior-first renumbering happens inside: keep the rank's own order)
    order0 = [np.arange(s.n) for s in states]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=80, device=0, max_coarse_size=10)
    # DistributedAMG renumbered the owned dofs [interior | boundary]; states[r].perm0 maps new -> bridge numbering
    bs = [torch.from_numpy(ref_owned[r][states[r].perm0]).cuda() for r in range(R)]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
    amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi").apply(np.concatenate([b.cpu().numpy() for b in bs]))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)
    # ---- CUMULATED solution back on every copy of a shared dof
    for r, vm in enumerate(vmaps):
        xo = np.empty(states[r].n)
        xo[states[r].perm0] = xs[r].cpu().numpy()
        vecs[r][:vm.n_own] = torch.from_numpy(xo).cuda()
    assert lib.amgx_halo_exchange(cc, R, hp, vp, 0) == 0, lib.amgx_comm_last_error(cc)
    lib.amgx_comm_synchronize(cc)
    xg = np.full(n, np.nan)
    for r, vm in enumerate(vmaps):
        xl = vm.from_ext(vecs[r].cpu().numpy(), locs[r].n_loc)
        prev = xg[gids[r]]
        assert np.all(np.isnan(prev) | (prev == xl))              # all copies of a shared dof agree
        xg[gids[r]] = xl
    assert not np.isnan(xg).any()
    for h in halos:
        lib.amgx_halo_destroy(h)
    lib.amgx_comm_destroy(cc)


@pytest.mark.parametrize("R,box,rot", [(2, (8, 7, 7), False), (4, (6, 6, 5), False), (2, (6, 6, 5), True), (8, (5, 5, 5), False)])
def test_loopback_device_elasticity_matches_serial_oracle(R, box, rot):
    """rank-partitioned elasticity levels on the device (3x3 -> 6x6 blocks, or 6x6 with rotations): halo pack kernels with
    block size 3 / 6, block-Jacobi in the literal stage order, replicated block tail; vs the serial oracle"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_elasticity_owned(r, pg, box, rotations=rot) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=10, device=0, max_coarse_size=5, energy=1, regularize_cmats=0 if rot else 1)
    bs0 = states[0].bs
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n * bs0) * np.repeat(s.free, bs0) for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n * bs0,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi").apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-11 * np.linalg.norm(ref)


@pytest.mark.parametrize("R,box,rot,stage_min", [(2, (8, 7, 7), False, 50), (4, (6, 6, 5), True, 50), (2, (9, 8, 8), True, 10 ** 9)])
def test_loopback_device_elasticity_hybrid_gs(R, box, rot, stage_min):
    """rank-partitioned elasticity levels with the hybrid block Gauss-Seidel smoother on the device: colour-range stages
    (local half / boundary / local half) around the halo exchange of x, block diagonal modified as in
    hybrid_smoother_utils.hpp:86-141; vs the oracle's serial hybrid GS (blocks = ranks)"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_elasticity_owned(r, pg, box, rotations=rot) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=10, device=0, max_coarse_size=5, energy=1, regularize_cmats=0 if rot else 1,
                           sm_type="gs", gs_stage_min_rows=stage_min)
    bs0 = states[0].bs
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n * bs0) * np.repeat(s.free, bs0) for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n * bs0,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg))
    ref = orc.apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
    assert orc.pcg(np.concatenate(bh), tol=1e-8, maxit=60)[1] < 45


@pytest.mark.parametrize("R,box,rot,B", [(2, (12, 9, 9), False, 42), (2, (12, 10, 10), True, 40), (4, (12, 12, 10), True, 20), (2, (16, 12, 12), False, None)])
def test_loopback_device_elasticity_block_hybrid_gs(R, box, rot, B):
    """sm_type = hgs on rank-partitioned ELASTICITY levels (3 x 3 / 6 x 6 blocks): bgsb_sweep_kernel over sweep blocks of B owned
    block rows -- boundary blocks, exchange of x, interior blocks; one-pass pre-smoothing with the `rest` image over
    [owned | ghost] columns; the block diagonal modified by every coupling that leaves a sweep block (other blocks of the rank and
    ghost columns alike) -- == the oracle's serial hybrid GS with the same blocks, colours and diagonal"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_elasticity_owned(r, pg, box, rotations=rot) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=150, device=0, max_coarse_size=5, energy=1, regularize_cmats=0 if rot else 1,
                           sm_type="hgs", **({"hgs_block_rows": B} if B else {}))
    assert amg.k >= 1 and all(s.gs_B > 0 for s in amg.dist_levels[0])
    bs0 = states[0].bs
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n * bs0) * np.repeat(s.free, bs0) for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n * bs0,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv))
    ref = orc.apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)
    assert orc.pcg(np.concatenate(bh), tol=1e-8, maxit=60)[1] < 45


def test_rccl_world_size_one_elasticity_hybrid_gs():
    """the same through a real RCCL communicator (world size 1, child process)"""
    _run_check("--world", "1", "--box", "10", "--elast", "6", "--sm", "gs", "--dmin", "50")


def test_rccl_world_size_one_elasticity_block_hybrid_gs():
    """sm_type = hgs on 6 x 6 block levels through a real RCCL communicator (world size 1, child process; graph replay included)"""
    _run_check("--world", "1", "--box", "14", "--elast", "6", "--sm", "hgs", "--dmin", "200")


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (8, (12, 12, 12), 3, 100), (4, (48, 48), 2, 300),
                                            (2, (40, 40, 40), 3, 20000)])          # last: the replicated tail has block-hybrid levels too
def test_loopback_device_block_hybrid_gs(R, box, dim, dmin):
    """sm_type = hgs on rank-partitioned levels through the native driver (boundary blocks, exchange, interior blocks; two
    launches per sweep) == the oracle's serial hybrid GS with the same blocks (256 owned rows), colours and modified diagonal"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type="hgs", hgs_block_rows=256)
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv)).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("sm", ["jacobi", "gs", "hgs"])
def test_loopback_device_rank_without_rows(sm):
    """a rank that owns nothing (NGSolve's classic MPI master) takes part in the native collective cycle with empty pieces:
    zero-row levels, halo tables without peers, an empty slot in the all-gather of level k"""
    import scipy.sparse as sp
    import torch
    from ngsamg_amd import bridge as B, dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_sm_types
    pgrid, gshape = (2, 2, 1), (17, 17, 9)
    R = 5
    comm = D.LoopbackComm(R)
    locs = []
    for r in range(R - 1):
        L, _ = B.shared_poisson_partition(r, pgrid, gshape)
        L.rank = r + 1
        L.dist_procs = [np.asarray(p) + 1 for p in L.dist_procs]
        locs.append(L)
    empty = B.SharedLocal(0, sp.csr_matrix((0, 0)), [], free=np.zeros(0, dtype=np.uint8), coords=np.zeros((0, 3)))
    states, vmaps = B.from_shared_layout(comm, [empty] + locs)
    assert states[0].n == 0
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=60, device=0, max_coarse_size=10, sm_type=sm, hgs_block_rows=256,
                           gs_stage_min_rows=50)
    rng = np.random.default_rng(1)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type=oracle_sm_types(amg)).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("R,box,dmin,sm", [(2, (14, 12, 12), 100, "jacobi"), (4, (10, 10, 10), 50, "jacobi"), (2, (14, 12, 12), 100, "gs"), (4, (10, 10, 10), 50, "hgs")])
def test_whole_cycle_graph_equals_direct_launches(R, box, dmin, sm, monkeypatch):
    """amgx_dist_apply captures the collective cycle (both streams, pack kernels, the wire) once per (b, x) and replays it:
    same result as direct launches, also for new contents of the same vectors and for a second pair of vectors"""
    import torch
    from ngsamg_amd import dist as D
    pg = D.proc_grid(R, 3)
    states = lambda: [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    rng = np.random.default_rng(3)

    def build(graph):
        monkeypatch.setenv("AMGX_DIST_GRAPH", "1" if graph else "0")
        return D.DistributedAMG(D.LoopbackComm(R), states(), dim=3, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm,
                                **({"hgs_block_rows": 256} if sm == "hgs" else {}))
    ag, ad = build(True), build(False)
    sts = ag.dist_levels[0]
    for rep in range(2):
        bs = [torch.from_numpy(rng.standard_normal(s.n) * s.free).cuda() for s in sts]
        xg = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in sts]
        xd = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in sts]
        for it in range(3):
            if it == 2:       # new right-hand side in the same storage: the captured graph must see it
                for b, s in zip(bs, sts):
                    b.copy_(torch.from_numpy(rng.standard_normal(s.n) * s.free))
            ag.Mult(bs, xg)
            ad.Mult(bs, xd)
            torch.cuda.synchronize()
            for a, b_ in zip(xg, xd):
                assert torch.equal(a, b_)
    gi, di = ag._dev.graph_info(), ad._dev.graph_info()
    assert gi["enabled"] and gi["graphs"] == 2 and gi["replays"] == 5, gi      # (the very first application runs directly: RCCL connections)
    assert not di["enabled"] and di["replays"] == 0
    assert ag._dev.n_exchanges() == ad._dev.n_exchanges()


@pytest.mark.parametrize("R,box,dmin,sm", [(2, (14, 12, 12), 100, "jacobi"), (4, (10, 10, 10), 50, "jacobi"), (3, (9, 12, 12), 80, "gs"), (2, (14, 12, 12), 100, "hgs")])
def test_distributed_pcg_history_equals_serial_oracle(R, box, dmin, sm):
    """amgx_dist_pcg (virtual ranks): owner-row SpMV with halo exchange, the collective cycle as preconditioner, global inner
    products; the error history equals the serial oracle's PCG on the assembled global hierarchy to 1e-6"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    pg = (R, 1, 1)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(D.LoopbackComm(R), states, dim=3, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm,
                           **({"hgs_block_rows": 256} if sm == "hgs" else {}))
    rng = np.random.default_rng(7)
    sts = amg.dist_levels[0]
    bh = [rng.standard_normal(s.n) * s.free for s in sts]
    bs = [torch.from_numpy(v).cuda() for v in bh]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in sts]
    it, errs = amg.pcg(bs, xs, tol=1e-8, maxsteps=100)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv))
    xo, ito, erro = orc.pcg(np.concatenate(bh), tol=1e-8, maxit=100)
    assert it == ito and it < 60
    erro = np.asarray(erro)[:it + 1]
    assert np.all(np.abs(errs - erro) <= 1e-6 * erro[0]) and np.allclose(errs[:5], erro[:5], rtol=1e-9)
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - xo) <= 1e-8 * np.linalg.norm(xo)
    # a second solve from the solution as initial guess stops immediately
    it2, errs2 = amg.pcg(bs, xs, tol=1e-8, maxsteps=100)
    assert errs2[0] <= 1e-7 * errs[0]


@pytest.mark.parametrize("pg,gshape,dmin,sm", [((3, 1, 1), (23, 14, 13), 100, "jacobi"), ((8, 1, 1), (43, 9, 10), 40, "jacobi"), ((5, 1, 1), (27, 12, 11), 60, "hgs")])
def test_strong_split_device_matches_serial_oracle(pg, gshape, dmin, sm):
    """bench.py --gpus N in miniature: ONE global grid cut into balanced, unequal slabs (virtual ranks on this GPU), two
    rank-partitioned levels, the gathered level, the collapsed replicated tail, the whole cycle replayed from its graph --
    against the serial oracle on the assembled global hierarchy, and the distributed PCG against the oracle's"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    R = int(np.prod(pg))
    states = [D.assemble_poisson_owned(r, pg, None, gshape=gshape, coords="rng") for r in range(R)]
    assert len({s.n for s in states}) > 1
    amg = D.DistributedAMG(D.LoopbackComm(R), states, dim=3, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm, spw=0,
                           **({"hgs_block_rows": 256} if sm == "hgs" else {}))
    sts = amg.dist_levels[0]
    rng = np.random.default_rng(5)
    bh = [rng.standard_normal(s.n) * s.free for s in sts]
    bs = [torch.from_numpy(v).cuda() for v in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in sts]
    for _ in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv))
    ref = orc.apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= (1e-12 if sm == "jacobi" else 1e-10) * np.linalg.norm(ref)
    assert amg._dev.graph_info()["replays"] == 1
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in sts]
    it, errs = amg.pcg(bs, xs, tol=1e-8, maxsteps=100)
    _, ito, erro = orc.pcg(np.concatenate(bh), tol=1e-8, maxit=100)
    assert it == ito and np.all(np.abs(errs - np.asarray(erro)[:it + 1]) <= 1e-6 * erro[0])


# ---- ProxySmoother (sm_steps / sm_symm) and the W-cycle on rank-partitioned levels: the native driver's step-by-step cycle ----

@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (4, (12, 12, 12), 3, 100), (4, (40, 40), 2, 100)])
@pytest.mark.parametrize("steps,symm,cycle", [(2, False, "V"), (1, True, "V"), (2, True, "V"), (1, False, "W"), (2, False, "W")])
def test_loopback_device_steps_symm_w_jacobi(R, box, dim, dmin, steps, symm, cycle):
    """reference: ProxySmoother (base_smoother.hpp:169-229) around the level smoother, AMGMatrix::SmoothW (amg_matrix.cpp:37-107);
    Jacobi steps of the parallel smoother with the plain diagonal are the global Jacobi steps, so the serial oracle with the same
    sm_steps / sm_symm / cycle on the assembled global hierarchy is the reference"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_steps=steps, sm_symm=symm, mg_cycle=cycle)
    assert amg.k >= 1 and not amg.fold
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(3):          # direct launches, capture, replay
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi", sm_steps=steps, sm_symm=symm, cycle=cycle).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (4, (48, 48), 2, 300)])
@pytest.mark.parametrize("steps,symm,cycle", [(2, False, "V"), (1, True, "V"), (1, False, "W")])
def test_loopback_device_steps_symm_w_block_hybrid_gs(R, box, dim, dmin, steps, symm, cycle):
    """the same for the block-hybrid Gauss-Seidel levels: every step is one hybrid sweep (exchange of x, then the sweep with the
    off-block values frozen), against the oracle's serial hybrid sweep with the same blocks, colours and modified diagonal"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, dim)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=dim, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type="hgs", hgs_block_rows=256,
                           sm_steps=steps, sm_symm=symm, mg_cycle=cycle)
    rng = np.random.default_rng(1)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(2):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv), sm_steps=steps, sm_symm=symm, cycle=cycle).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-10 * np.linalg.norm(ref)


@pytest.mark.parametrize("R,box,dmin,sm,restart", [(2, (14, 12, 12), 100, "jacobi", 30), (4, (10, 10, 10), 50, "jacobi", 5), (2, (14, 12, 12), 100, "hgs", 12)])
def test_distributed_gmres_history_equals_serial_oracle(R, box, dmin, sm, restart):
    """amgx_dist_gmres (rank-local Arnoldi vectors, fused inner products + all-reduce) against the oracle's serial GMRES on the
    assembled global hierarchy: same iteration count, same error history, same solution"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm,
                           **({"hgs_block_rows": 256} if sm == "hgs" else {}))
    rng = np.random.default_rng(3)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
    it, errs = amg.gmres(bs, xs, tol=1e-9, maxsteps=150, restart=restart)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type="jacobi") if sm == "jacobi" else Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv))
    xo, ito, erro = orc.gmres(np.concatenate(bh), tol=1e-9, maxit=150, restart=restart)
    assert it == ito
    assert errs.shape == erro.shape and np.all(np.abs(errs - erro) <= 1e-6 * erro[0])
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - xo) <= 1e-7 * np.linalg.norm(xo)
    # a second solve re-uses the workspace (and the preconditioner's captured graph)
    for x in xs:
        x.zero_()
    it2, errs2 = amg.gmres(bs, xs, tol=1e-9, maxsteps=150, restart=restart)
    assert it2 == it and np.array_equal(errs2, errs)


@pytest.mark.parametrize("R,box,dmin,sm", [(2, (14, 12, 12), 100, "jacobi"), (4, (10, 10, 10), 50, "hgs")])
def test_distributed_single_reduction_pcg_equals_serial_oracle(R, box, dmin, sm):
    """amgx_dist_pcg with AMGX_PCG_SINGLE_REDUCTION: one all-reduce of (gamma, delta) per iteration; history of the serial oracle's
    classical PCG on the assembled global hierarchy to 1e-6"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=dmin, device=0, max_coarse_size=10, sm_type=sm,
                           **({"hgs_block_rows": 256} if sm == "hgs" else {}))
    rng = np.random.default_rng(5)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
    it, errs = amg.pcg(bs, xs, tol=1e-9, maxsteps=100, single_reduction=True)
    torch.cuda.synchronize()
    glv = amg.global_levels()
    orc = Oracle(glv, sm_type="jacobi") if sm == "jacobi" else Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv))
    xo, ito, erro = orc.pcg(np.concatenate(bh), tol=1e-9, maxit=100)
    assert abs(it - ito) <= 1
    k = min(it, ito)
    assert np.allclose(errs[:k], erro[:k], rtol=1e-6)
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - xo) <= 1e-7 * np.linalg.norm(xo)


@pytest.mark.parametrize("R,box,dmin", [(2, (20, 18, 16), 150), (4, (14, 14, 12), 40)])
def test_loopback_device_local_window_image_on_rank_partitioned_levels(R, box, dmin, monkeypatch):
    """the local-window image of A' (gathered vector staged in LDS, sell_lw_pre_restrict_kernel) on rank-partitioned coarse levels:
    interior chunks run beside the exchange (their windows hold owned columns only), boundary chunks after it (ghost columns are
    ordinary columns of the [owned | ghost] vector) -- forced onto the small level 1 of this case"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_LW_MIN_ROWS", "50")
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    states = [D.assemble_poisson_owned(r, pg, box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=dmin, device=0, max_coarse_size=10)
    assert amg.k >= 2, "the case needs a rank-partitioned level 1"
    assert any(op.top.matrix_info(1, "ApreLW")["fmt"] == "sell-lw" for op in amg.ops), "no rank took the local-window image on level 1"
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.full((s.n,), float("nan"), dtype=torch.float64, device="cuda") for s in states]
    for rep in range(3):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    ref = Oracle(amg.global_levels(), sm_type="jacobi").apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    assert np.linalg.norm(got - ref) <= 1e-12 * np.linalg.norm(ref)
