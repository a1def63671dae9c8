#!/usr/bin/env python3
"""Golden fixtures for the rank-partitioned (hybrid) smoothers on synthetic partitions ("hybrid-GS with a
2x2 and 2x2x2 synthetic partition"): the GLOBAL hierarchy assembled from all ranks (level matrices, P, modified
diagonals, rank of every row, per-rank colour-major visiting order) and the result of one V-cycle of the serial hybrid
oracle, for Jacobi, hybrid Gauss-Seidel and hybrid block Gauss-Seidel.  Generated with virtual ranks (LoopbackComm) and
the CPU stage backend; pins the distributed setup (partition, aggregates, halo tables) and both execution paths.

Run from the repo root:  python tests/golden/make_golden_hybrid.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from ngsamg_amd import dist as D                      # noqa: E402
from oracle.pyoracle import Oracle                    # noqa: E402
from tests.dist_cpu_backend import cpu_backend        # noqa: E402
from tests.dist_oracle import oracle_bgs, oracle_sm_types   # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CASES = {"hybrid_poisson2d_2x2": dict(R=4, box=(9, 9), dim=2, dmin=30),
         "hybrid_poisson3d_2x2x2": dict(R=8, box=(5, 5, 5), dim=3, dmin=20)}


def run(case, sm):
    comm = D.LoopbackComm(case["R"])
    pg = D.proc_grid(case["R"], case["dim"])
    states = [D.assemble_poisson_owned(r, pg, case["box"]) for r in range(case["R"])]
    amg = D.DistributedAMG(comm, states, dim=case["dim"], dist_min_rows=case["dmin"], backend=cpu_backend(sm_type=sm),
                           max_coarse_size=10, sm_type=sm)
    rng = np.random.default_rng(0)
    bs = [torch.from_numpy(rng.standard_normal(s.n) * s.free) for s in states]
    xs = [torch.zeros(s.n, dtype=torch.float64) for s in states]
    amg.Mult(bs, xs)
    return amg, np.concatenate([b.numpy() for b in bs]), np.concatenate([x.numpy() for x in xs])


def main():
    for name, case in CASES.items():
        d = {}
        for sm in ("jacobi", "gs", "bgs"):
            amg, b, x = run(case, sm)
            glv = amg.global_levels()
            ref = Oracle(glv, sm_type=oracle_sm_types(amg), bgs=oracle_bgs(amg, glv)).apply(b)
            assert np.linalg.norm(x - ref) <= 1e-12 * np.linalg.norm(ref)
            d[f"{sm}_V"] = ref
            d[f"{sm}_k"] = np.int64(amg.k)
            if sm == "jacobi":
                d["b"] = b
                d["n_levels"] = np.int64(len(glv))
                d["rank_sizes"] = np.array([s.n for s in amg.dist_levels[0]], dtype=np.int64)
                for l, L in enumerate(glv):
                    for tag, M in (("A", L.A), ("P", L.P), ("PT", L.PT)):
                        if M is None:
                            continue
                        d[f"l{l}_{tag}_shape"] = np.array([M.n_rows, M.n_cols, M.br, M.bc], dtype=np.int64)
                        d[f"l{l}_{tag}_rowptr"], d[f"l{l}_{tag}_col"], d[f"l{l}_{tag}_val"] = M.rowptr.copy(), M.col.copy(), M.val.copy()
                    d[f"l{l}_free"] = np.asarray(L.free).copy()
                    d[f"l{l}_dinv"] = np.asarray(L.dinv).copy()
                    d[f"l{l}_color"] = np.asarray(L.color).copy()
            else:
                for l, L in enumerate(glv[: amg.k]):
                    d[f"{sm}_l{l}_dinv"] = np.asarray(L.dinv).copy()
                    d[f"{sm}_l{l}_gs_block"] = np.asarray(L.gs_block).copy()
                    if sm == "gs":
                        d[f"{sm}_l{l}_gs_order"] = np.asarray(L.gs_order).copy()
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "ranks", case["R"], "global n", d["b"].size, "distributed levels", int(d["jacobi_k"]))


if __name__ == "__main__":
    main()
