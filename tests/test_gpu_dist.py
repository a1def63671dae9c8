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
    ref = Oracle(glv, sm_type=amg.oracle_sm_types(), bgs=amg.oracle_bgs(glv)).apply(np.concatenate(bh))
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


def test_rccl_point_to_point_self_loop():
    """the RCCL branch of TorchComm.halo (batch_isend_irecv on device tensors, stream-ordered wait) on ONE GPU: world size
    1, the rank sends to and receives from itself; run in a child process (its own process group)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(29600 + os.getpid() % 300))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_selfloop.py")], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "correct = True" in r.stdout


@pytest.mark.parametrize("R,box,dim,dmin", [(2, (16, 16, 16), 3, 200), (4, (12, 12, 12), 3, 100), (3, (20, 9, 9), 3, 100)])
def test_loopback_device_fused_kernels_on_rank_partitioned_levels(R, box, dim, dmin, monkeypatch):
    """the production shape of the multi-GPU run: level 0 in the one-thread-per-row form, i.e. the FUSED down kernel
    (pre-smoothing + chunk-local restriction, omega*Dinv in the diagonal slot) and the windowed Q kernel on matrices WITH
    ghost columns -- forced onto small levels"""
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    monkeypatch.setenv("AMGX_SELL_MAX_LANES", "1")
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
