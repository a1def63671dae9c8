#!/usr/bin/env python3
"""Rank-partitioned V-cycle over RCCL through the C ABI (amgx_comm_* / amgx_dist_* / amgx_halo_*), one process per GPU,
checked against the serial CPU oracle on the assembled global hierarchy.

    python tests/dist_rccl_check.py [--world N] [--box B] [--sm jacobi|gs|bgs] [--pgrid slab|box] [--no-fold]

Without RANK in the environment the script starts its N ranks itself (fresh child processes, before any GPU call) and
relays rank 0's verdict.  World size 1 runs on a one-GPU box: RCCL initialisation, all-gather and the whole native
driver run; the point-to-point wire is exercised by a halo map whose only peer is the rank itself (self send / receive).
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--box", type=int, default=24)
    ap.add_argument("--sm", default="jacobi")
    ap.add_argument("--pgrid", default="box", choices=["box", "slab"])
    ap.add_argument("--no-fold", action="store_true")
    ap.add_argument("--elast", default="", choices=["", "3", "6"], help="linear elasticity with 3x3 (displacements) or 6x6 (with rotations) fine blocks")
    ap.add_argument("--dmin", type=int, default=500)
    ap.add_argument("--port", type=int, default=0)
    return ap.parse_args()


def launch(args):
    import socket
    port = args.port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    import signal
    import bench                       # the rank watchdog is the one bench.py --gpus N uses
    procs = []

    def reap(*_):
        for p in procs:                # exactly the rank processes started here
            if p.poll() is None:
                p.kill()

    # a parent that is told to stop (pytest's subprocess timeout sends SIGKILL to us only if we ignore SIGTERM; a driver
    # usually sends SIGTERM first) must not leave ranks behind that hold the GPUs
    signal.signal(signal.SIGTERM, lambda *a: (reap(), sys.exit(143)))
    try:
        for r in range(args.world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS=str(max(1, (os.cpu_count() or 8) // args.world)))
            # own session: a rank that outlives this launcher would otherwise keep the caller's process group alive
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, start_new_session=True,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
        # first non-zero exit code, or no verdict within the limit: the remaining ranks are killed (a rank that died in RCCL /
        # gloo initialisation or a collective that never completes would otherwise block the others forever)
        codes, out, why = bench._wait_ranks(procs, float(os.environ.get("NGSAMG_CHECK_TIMEOUT", "600")))
    finally:
        reap()
    sys.stdout.write(out or "")
    if why:
        print("stopped:", why)
    print("exit codes", codes)
    sys.exit(0 if all(c == 0 for c in codes) and not why else 1)


def self_loop_halo(lib, _lib, comm, torch, dev):
    """world size 1: a halo map whose peer is the rank itself -- pack kernel, ncclSend / ncclRecv to self inside one
    group, receive straight into the ghost segment; then the add direction"""
    n, ng, bs = 5000, 700, 3
    rng = np.random.default_rng(3)
    idx = np.sort(rng.choice(n, ng, replace=False)).astype(np.int32)
    peers = np.zeros(1, dtype=np.int32)
    sp = np.array([0, ng], dtype=np.int64)
    d = _lib.amgx_halo_desc()
    d.n_peers = 1
    d.peer_rank, d.send_ptr, d.send_idx, d.recv_ptr = _lib.ptr(peers, C.c_int32), _lib.ptr(sp, C.c_int64), _lib.ptr(idx, C.c_int32), _lib.ptr(sp, C.c_int64)
    h = C.c_void_p()
    assert lib.amgx_halo_create(comm, C.byref(d), n, ng, bs, 0, C.byref(h)) == 0, lib.amgx_comm_last_error(comm)
    v0 = rng.standard_normal((n + ng) * bs)
    v = torch.from_numpy(v0.copy()).to(dev)
    hp, vp = (C.c_void_p * 1)(h), (C.c_void_p * 1)(v.data_ptr())
    assert lib.amgx_halo_exchange(comm, 1, hp, vp, 0) == 0, lib.amgx_comm_last_error(comm)
    lib.amgx_comm_synchronize(comm)
    got = v.cpu().numpy().reshape(-1, bs)
    exp = v0.reshape(-1, bs).copy()
    exp[n:] = exp[idx]
    ok0 = np.array_equal(got, exp)
    assert lib.amgx_halo_exchange(comm, 1, hp, vp, 1) == 0, lib.amgx_comm_last_error(comm)
    lib.amgx_comm_synchronize(comm)
    got = v.cpu().numpy().reshape(-1, bs)
    exp2 = exp.copy()
    exp2[idx] += exp[n:]
    exp2[n:] = 0.0
    ok1 = np.allclose(got, exp2, rtol=0, atol=0)
    lib.amgx_halo_destroy(h)
    return ok0 and ok1


def main():
    args = parse()
    if "RANK" not in os.environ:
        if args.world > 1:
            import torch
            if torch.cuda.device_count() < args.world:          # (counting devices does not initialise the GPU)
                print(f"need {args.world} GPUs, found {torch.cuda.device_count()}")
                sys.exit(2)
        launch(args)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    from ngsamg_amd import _lib, dist as D
    from oracle.pyoracle import Oracle
    from tests.dist_oracle import oracle_bgs, oracle_sm_types
    dev = int(os.environ.get("LOCAL_RANK", rank)) % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # bootstrap + host-side setup messages only
    try:
        comm = D.TorchComm()
        pg = (world, 1, 1) if args.pgrid == "slab" else D.proc_grid(world, 3)
        if args.elast:
            st = D.assemble_elasticity_owned(rank, pg, (args.box,) * 3, rotations=args.elast == "6")
            amg = D.DistributedAMG(comm, [st], dim=3, dist_min_rows=args.dmin, device=dev, max_coarse_size=10, energy=1,
                                   regularize_cmats=0 if args.elast == "6" else 1, sm_type=args.sm, gs_stage_min_rows=100)
        else:
            st = D.assemble_poisson_owned(rank, pg, (args.box,) * 3)
            amg = D.DistributedAMG(comm, [st], dim=3, dist_min_rows=args.dmin, device=dev, max_coarse_size=20, sm_type=args.sm,
                                   fold=not args.no_fold, gs_stage_min_rows=1000)
        bs0 = getattr(st, "bs", 1)
        assert amg._dev is not None
        lib = _lib.hip()
        kind, nr, rk = C.c_int32(), C.c_int32(), C.c_int32()
        lib.amgx_comm_info(amg._dev._comm, C.byref(kind), C.byref(nr), C.byref(rk), None)
        assert (kind.value, nr.value, rk.value) == (_lib.AMGX_COMM_RCCL, world, rank), "RCCL communicator does not span all ranks"
        rng = np.random.default_rng(rank)
        bh = rng.standard_normal(st.n * bs0) * np.repeat(st.free, bs0)
        b = torch.from_numpy(bh).to(f"cuda:{dev}")
        x = torch.full_like(b, float("nan"))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                amg.Mult([b], [x])
        torch.cuda.synchronize()
        gi = amg._dev.graph_info()
        loop_ok = self_loop_halo(lib, _lib, amg._dev._comm, torch, f"cuda:{dev}") if world == 1 else True
        glv = amg.global_levels()
        allb, allx = [None] * world, [None] * world
        dist.all_gather_object(allb, bh)
        dist.all_gather_object(allx, x.cpu().numpy())
        if rank == 0:
            ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv)).apply(np.concatenate(allb))
            got = np.concatenate(allx)
            err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            tol = (1e-12 if args.sm == "jacobi" else 1e-10) if not args.elast else (1e-11 if args.sm == "jacobi" else 1e-10)
            print(f"world={world} pgrid={pg} box={args.box}^3 sm={args.sm} elast={args.elast or 'no'} fold={amg.fold} distributed levels={amg.k} "
                  f"exchanges per cycle={amg._dev.n_exchanges() // 3} rel.err vs serial oracle = {err:.3e} self-loop halo ok = {loop_ok} "
                  f"graph enabled = {gi['enabled']} graphs = {gi['graphs']} replays = {gi['replays']} note = '{gi['note']}' "
                  f"allgather = {os.environ.get('AMGX_DIST_FORCE_ALLGATHER', 'default')}")
            ok = err < tol and loop_ok
            print("RCCL CHECK", "PASSED" if ok else "FAILED")
            code = 0 if ok else 1
        else:
            code = 0
    finally:
        dist.destroy_process_group()
    sys.exit(code)


if __name__ == "__main__":
    main()
