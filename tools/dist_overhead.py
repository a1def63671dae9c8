#!/usr/bin/env python3
"""Overhead of the rank-partitioned cycle driver vs the single-GPU graph replay: R virtual ranks (LoopbackComm) on ONE
GPU at full box size; kernels of the ranks serialise on the device, so (time / R) - single-GPU cycle time = per-rank
cost of staging (Python + torch index_select + copies), i.e. what a real multi-GPU run adds on top of the RCCL time.
python tools/dist_overhead.py [nv] [R]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ngsamg_amd import dist as D
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 215
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    sm = sys.argv[3] if len(sys.argv) > 3 else "jacobi"
    comm = D.LoopbackComm(R)
    pg = D.proc_grid(R, 3)
    t0 = time.time()
    states = [D.assemble_poisson_owned(r, pg, (nv, nv, nv)) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=50000, device=0, max_coarse_size=50, sm_type=sm)
    print(f"setup {time.time() - t0:.1f}s, distributed levels {amg.k}, sizes {[lv[0].n for lv in amg.dist_levels]}, "
          f"ghosts {[lv[0].ghost_owner.size for lv in amg.dist_levels]}, tail n = {amg.tail_hier.levels[0].n}")
    rng = np.random.default_rng(0)
    bs = [amg.rhs_buffer(i) if sm == 'jacobi' else torch.zeros(s.n, dtype=torch.float64, device='cuda') for i, s in enumerate(states)]
    for b, s in zip(bs, states):
        b.copy_(torch.from_numpy(rng.standard_normal(s.n) * s.free))
    xs = [torch.zeros_like(b) for b in bs]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            amg.Mult(bs, xs)
        torch.cuda.synchronize()
        K = 30
        t0 = time.perf_counter()
        for _ in range(K):
            amg.Mult(bs, xs)
        t_host = time.perf_counter() - t0          # host enqueue time
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
    print(f"R={R} nv={nv} {sm}: {dt * 1e3:.3f} ms per distributed cycle of all ranks = {dt * 1e3 / R:.3f} ms per rank; "
          f"host enqueue {t_host / K * 1e3:.3f} ms per cycle")
    single = amg.ops[0].top.time_op(0, 0, 20)
    print(f"level-0 residual kernel on rank 0: {single * 1e3:.1f} us")


if __name__ == "__main__":
    main()
