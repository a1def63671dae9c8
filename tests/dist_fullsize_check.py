#!/usr/bin/env python3
"""One-off check of the multi-GPU production shape at full size on ONE GPU: two virtual ranks (LoopbackComm), slab
partition, 2 x (108 x 215 x 215) vertices = 10 M DOF: level 0 in the one-thread-per-row formats with ghost columns (fused
down kernel, windowed Q), result against the serial oracle on the assembled global hierarchy."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root


def main():
    import torch
    from ngsamg_amd import dist as D
    from oracle.pyoracle import Oracle
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    box = tuple(int(v) for v in sys.argv[2].split("x")) if len(sys.argv) > 2 else (108, 215, 215)
    check = (sys.argv[3] != "nocheck") if len(sys.argv) > 3 else True
    comm = D.LoopbackComm(R)
    t0 = time.time()
    states = [D.assemble_poisson_owned(r, (R, 1, 1), box) for r in range(R)]
    amg = D.DistributedAMG(comm, states, dim=3, dist_min_rows=50000, device=0, max_coarse_size=50, max_levels=10)
    print(f"setup {time.time() - t0:.1f}s; distributed levels {amg.k}, sizes {[lv[0].n for lv in amg.dist_levels]}, "
          f"ghosts {[lv[0].ghost_owner.size for lv in amg.dist_levels]}", flush=True)
    top = amg.ops[0].top
    for w in ("A", "Apre", "P", "PT", "Q"):
        print(" level 0", w, top.matrix_info(0, w), flush=True)
    rng = np.random.default_rng(0)
    bh = [rng.standard_normal(s.n) * s.free for s in states]
    bs = [torch.from_numpy(b).cuda() for b in bh]
    xs = [torch.zeros(s.n, dtype=torch.float64, device="cuda") for s in states]
    for _ in range(3):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        amg.Mult(bs, xs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20 * 1e3
    print(f"cycle of all {R} virtual ranks (box {box}): {dt:.3f} ms = {dt / R:.3f} ms per rank; exchanges per cycle {amg._dev.n_exchanges() // 23}", flush=True)
    if not check:
        return
    glv = amg.global_levels()
    ref = Oracle(glv, sm_type="jacobi", threads=16).apply(np.concatenate(bh))
    got = np.concatenate([x.cpu().numpy() for x in xs])
    err = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(f"rel. error vs serial oracle on the global hierarchy: {err:.3e}")
    sys.exit(0 if err < 1e-12 else 1)


if __name__ == "__main__":
    main()
