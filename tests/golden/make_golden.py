#!/usr/bin/env python3
"""Generates the committed golden fixtures tests/golden/*.npz.

The reference holds no golden vectors for the apply path and cannot be run here, so these
fixtures are produced by the repo's own host setup (hierarchy arrays) and CPU oracle (stage vectors of one
SmoothV, PCG residual histories) on tiny seeded inputs, as SURVEY.md 8c "fixtures the build should
therefore create itself" prescribes.  They pin the oracle and the GPU path against regressions; they do
not pin either to the reference ("parity unpinned").

Each fixture stores every level's CSR arrays, P, PT, dinv, free mask, colours, and for each smoother kind
the vectors after each stage of one V-cycle:  x after pre-smoothing, r, b_1 (level 0), the final x,
plus W / BS cycle outputs and the PCG error history.

Run from the repo root:  python tests/golden/make_golden.py [fixture names; default all]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from ngsamg_amd import fem                       # noqa: E402
from ngsamg_amd._lib import Matrix               # noqa: E402
from ngsamg_amd.hierarchy import Hierarchy       # noqa: E402
from oracle.pyoracle import Oracle               # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "poisson2d_9": dict(kind="poisson", shape=(9, 9), diri="left|top", mcs=5),
    "poisson2d_17": dict(kind="poisson", shape=(17, 17), diri="left|top", mcs=5),
    "poisson3d_5": dict(kind="poisson", shape=(5, 5, 5), diri="right|top", mcs=5),
    "poisson3d_9": dict(kind="poisson", shape=(9, 9, 9), diri="right|top", mcs=10),
    "elast3d_4_bs3": dict(kind="elast", shape=(4, 4, 4), diri="left", mcs=4, rot=False),
    "elast3d_4_bs6": dict(kind="elast", shape=(4, 4, 4), diri="left", mcs=4, rot=True),
    # hierarchies of the edge-matrix setup: general 3x6 / 6x6 blocks in P instead of w Q(t)
    "elast3d_5_bs3_edge_mats": dict(kind="elast", shape=(5, 4, 4), diri="left", mcs=4, rot=False, extra={"edge_mats": 1}),
    "elast3d_4_bs6_edge_mats": dict(kind="elast", shape=(4, 4, 4), diri="left", mcs=4, rot=True,
                                    extra={"edge_mats": 1, "crs_robust": 1, "sp_improve_its": 1}),
}


def build(case):
    if case["kind"] == "poisson":
        p = fem.poisson_fast(case["shape"], dirichlet=case["diri"])
        energy = 0
        kw = {}
    else:
        p = fem.elasticity_fast(case["shape"], dirichlet=case["diri"], mu=1.0, lam=0.5, rotations=case["rot"])
        energy = 1
        kw = {"regularize_cmats": 0 if case["rot"] else 1}
        kw.update(case.get("extra", {}))
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, dim=len(case["shape"]), energy=energy, max_coarse_size=case["mcs"], **kw)
    return p, H


def main():
    only = sys.argv[1:]          # python tests/golden/make_golden.py [names ...]: regenerate just these
    for name, case in CASES.items():
        if only and name not in only:
            continue
        p, H = build(case)
        d = {"n_levels": np.int64(H.n_levels), "bs0": np.int64(p.bs)}
        for l, L in enumerate(H.levels):
            for tag, M in (("A", L.A), ("P", L.P), ("PT", L.PT)):
                if M is None:
                    continue
                d[f"l{l}_{tag}_shape"] = np.array([M.n_rows, M.n_cols, M.br, M.bc], dtype=np.int64)
                d[f"l{l}_{tag}_rowptr"] = M.rowptr.copy()
                d[f"l{l}_{tag}_col"] = M.col.copy()
                d[f"l{l}_{tag}_val"] = M.val.copy()
            d[f"l{l}_free"] = L.free.copy()
            d[f"l{l}_dinv"] = L.dinv.copy()
            d[f"l{l}_color"] = L.color.copy()
        rng = np.random.default_rng(0)
        b = rng.standard_normal(p.n * p.bs) * np.repeat(p.free, p.bs)
        d["b"] = b
        d["load"] = np.asarray(p.load, dtype=np.float64).copy()      # right-hand side of the PCG histories below
        for sm in ("jacobi", "gs", "gs_mc"):
            orc = Oracle(H.levels, sm_type=sm)
            x = np.zeros_like(b)
            r = b.copy()
            orc.smooth(0, x, b, r, True, True, True)
            d[f"{sm}_x_pre"] = x.copy()
            d[f"{sm}_r_pre"] = r.copy()
            if H.n_levels > 1:
                d[f"{sm}_b1"] = orc.transfer_f2c(0, r)
            d[f"{sm}_V"] = orc.apply(b)
            d[f"{sm}_W"] = Oracle(H.levels, sm_type=sm, cycle="W").apply(b)
            d[f"{sm}_BS"] = Oracle(H.levels, sm_type=sm, cycle="BS").apply(b)
            d[f"{sm}_V_symm2"] = Oracle(H.levels, sm_type=sm, sm_steps=2, sm_symm=True).apply(b)
            _, it, errs = orc.pcg(p.load, tol=1e-8, maxit=100)
            d[f"{sm}_pcg_errs"] = errs
        # block Gauss-Seidel over the aggregates (reference BSmoother): block tables, inverted diagonal blocks, block
        # colours of every smoothed level + the same stage / cycle vectors ("bgs" natural, "bgs_mc" colour-major order)
        bgs = H.build_bgs()
        for l, g in enumerate(bgs[:-1]):
            d[f"l{l}_bgs_block_ptr"] = g.block_ptr.copy()
            d[f"l{l}_bgs_block_rows"] = g.block_rows.copy()
            d[f"l{l}_bgs_dinv_ptr"] = g.dinv_ptr.copy()
            d[f"l{l}_bgs_dinv"] = g.dinv[: int(g.dinv_ptr[-1])].copy()
            d[f"l{l}_bgs_color"] = g.color.copy()
        for sm in ("bgs", "bgs_mc"):
            orc = Oracle(H.levels, sm_type=sm, bgs=bgs)
            x = np.zeros_like(b)
            r = b.copy()
            orc.smooth(0, x, b, r, True, True, True)
            d[f"{sm}_x_pre"] = x.copy()
            d[f"{sm}_r_pre"] = r.copy()
            d[f"{sm}_V"] = orc.apply(b)
            d[f"{sm}_W"] = Oracle(H.levels, sm_type=sm, cycle="W", bgs=bgs).apply(b)
            _, it, errs = orc.pcg(p.load, tol=1e-8, maxit=100)
            d[f"{sm}_pcg_errs"] = errs
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        print(name, "levels", H.n_levels, "n", p.n, "bs", p.bs)


if __name__ == "__main__":
    main()
