#!/usr/bin/env python3
"""Side measurement (not the driver's bench contract): block-CSR V-cycle on the elasticity configs
(SURVEY.md 8d cfg 3: 3x3 fine / 6x6 coarse, cfg 5: 6x6 everywhere).  python tools/bench_block.py [nv] [rot]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from ngsamg_amd import fem, Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix, vcycle_bytes, matrix_bytes
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rot = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    sm = sys.argv[3] if len(sys.argv) > 3 else "jacobi"
    t0 = time.time()
    p = fem.elasticity_fast((nv, nv, nv), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    t1 = time.time()
    H = Hierarchy(A, p.free, p.coords, dim=3, energy=1, max_coarse_size=50, regularize_cmats=0 if rot else 1)
    if sm == "bgs":
        H.build_bgs()
    t2 = time.time()
    amg = DeviceAMGMatrix(H, sm_type=sm, device=0)
    t3 = time.time()
    print(f"assembly {t1 - t0:.1f}s hierarchy {t2 - t1:.1f}s upload {t3 - t2:.1f}s", file=sys.stderr)
    print(H.summary(), file=sys.stderr)
    total, per = vcycle_bytes(H)
    rng = np.random.default_rng(0)
    b = torch.from_numpy(rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)).cuda()
    x = torch.empty_like(b)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(5):
            amg.Mult(b, x)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        K = 50
        for _ in range(K):
            amg.Mult(b, x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - ts) / K
    out = {"workload": f"3D elasticity {nv}^3 nodes, bs={p.bs}, rot={rot}, {sm}", "applies_per_s": 1 / dt, "ms": dt * 1e3,
           "cycle_bytes": total, "GBs_algorithmic": total / dt / 1e9, "levels": [(L.n, L.bs, L.A.nnz) for L in H.levels]}
    for l in range(min(2, H.n_levels - 1)):
        for op, nm in ((0, "res"), (1, "jac"), (2, "restrict"), (3, "prolong")):
            if op == 1 and sm != "jacobi":
                continue
            ms = amg.time_op(l, op, 20)
            M = H.levels[l].A if op < 2 else (H.levels[l].PT if op == 2 else H.levels[l].P)
            by = matrix_bytes(M) + 8 * H.levels[l].n * H.levels[l].bs * (3 if op == 0 else 4 if op == 1 else 1 if op == 2 else 2)
            out[f"l{l}_{nm}_us"] = ms * 1e3
            out[f"l{l}_{nm}_GBs"] = by / ms / 1e6
    print(json.dumps(out))


if __name__ == "__main__":
    main()
