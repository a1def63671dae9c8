#!/usr/bin/env python3
"""Convergence probe (CPU oracle only): PCG iterations of the V(1,1) cycle with
  gs      sequential Gauss-Seidel in natural order          (reference GSS3, gssmoother.cpp:196-315)
  gs_mc   multicolour order                                 (round-1 GPU kernels)
  gs_blk  block-hybrid: blocks of B consecutive rows, GS inside a block (colour-major), couplings to other blocks
          frozen at their sweep-start values, l1-modified diagonal  (reference HybridGSSmoother with blocks = "ranks",
          hybrid_smoother_utils.hpp:111-142)
Usage: python tests/gs_block_probe.py NV [B ...]"""
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def block_hybrid_levels(levels, B):
    from copy import copy
    out = []
    for i, lv in enumerate(levels):
        L = copy(lv)
        if i + 1 < len(levels):
            A = lv.A.to_scipy().tocsr()
            n = A.shape[0]
            blk = (np.arange(n) // B).astype(np.int32)
            d = A.diagonal()
            coo = A.tocoo()
            off = blk[coo.row] != blk[coo.col]
            sd = np.sqrt(np.where(d > 0, d, 1.0))
            ad = np.zeros(n)
            np.add.at(ad, coo.row[off], np.abs(coo.data[off]) / (sd[coo.row[off]] * sd[coo.col[off]]))
            md = np.maximum(1.0, 0.51 * (1.0 + ad)) * d
            free = np.asarray(lv.free).astype(bool)
            L.dinv = np.where(free & (md != 0), 1.0 / np.where(md != 0, md, 1.0), 0.0)
            color = np.asarray(lv.color)
            rows = np.nonzero(color >= 0)[0]
            key = blk[rows].astype(np.int64) * (color.max() + 2) + color[rows]
            L.gs_order = rows[np.argsort(key, kind="stable")].astype(np.int32)
            L.gs_block = blk
            L.frac_mod = float(np.mean(md[free] > d[free] * (1 + 1e-12)))
        out.append(L)
    return out


def main():
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from oracle.pyoracle import Oracle
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    Bs = [int(v) for v in sys.argv[2:]] or [256, 1024, 4096]
    p = fem.poisson_fast((nv, nv, nv), dirichlet="right|top", jitter=0.2, seed=1)
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10)
    print(H.summary(), flush=True)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    for tol in (1e-8, 1e-12):
        res = {}
        for sm in ("jacobi", "gs", "gs_mc"):
            _, it, _ = Oracle(H.levels, sm_type=sm, threads=8).pcg(b, tol=tol, maxit=300)
            res[sm] = it
        for B in Bs:
            lv = block_hybrid_levels(H.levels, B)
            types = ["gs_order"] * (len(lv) - 1) + ["gs_mc"]
            _, it, _ = Oracle(lv, sm_type=types, threads=8).pcg(b, tol=tol, maxit=300)
            res[f"gs_blk{B}"] = (it, round(lv[0].frac_mod, 3))
        print(f"nv={nv} tol={tol:g}: {res}", flush=True)


if __name__ == "__main__":
    main()
