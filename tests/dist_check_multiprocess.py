#!/usr/bin/env python3
"""Multi-PROCESS check of the rank-partitioned V-cycle on ONE GPU: W processes share cuda:0, transport = gloo with
host-staged halos (debug transport; RCCL refuses several ranks per device).  Exercises everything of the N > 1 path
except the RCCL calls themselves.   python tests/dist_check_multiprocess.py [W] [box]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
sys.path.insert(0, ROOT)


def worker(rank, world, port, box, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS="8")
    import torch
    import torch.distributed as dist
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = D.TorchComm()
        pg = D.proc_grid(world, 3)
        st = D.assemble_poisson_owned(rank, pg, (box, box, box))
        amg = D.DistributedAMG(comm, [st], dim=3, dist_min_rows=500, device=0, max_coarse_size=20)
        rng = np.random.default_rng(rank)
        bh = rng.standard_normal(st.n) * st.free
        b = torch.from_numpy(bh).cuda()
        x = torch.zeros_like(b)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(3):
                amg.Mult([b], [x])
        torch.cuda.synchronize()
        glv = amg.global_levels()
        allb, allx = [None] * world, [None] * world
        dist.all_gather_object(allb, bh)
        dist.all_gather_object(allx, x.cpu().numpy())
        if rank == 0:
            ref = Oracle(glv, sm_type="jacobi").apply(np.concatenate(allb))
            got = np.concatenate(allx)
            q.put((amg.k, float(np.linalg.norm(got - ref) / np.linalg.norm(ref))))
    finally:
        dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    box = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, W, 29611, box, q)) for r in range(W)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
    codes = [p.exitcode for p in procs]
    print("exit codes", codes)
    k, err = q.get(timeout=10)
    print(f"W={W} box={box}^3 distributed levels={k} rel.err vs serial oracle = {err:.3e}")
    assert all(c == 0 for c in codes) and err < 1e-12


if __name__ == "__main__":
    main()
