#!/usr/bin/env python3
"""Counterpart of the reference's examples/elasticity/simple.py on this build's stand-in FEM assembly (NGSolve / netgen
are not available): linear elasticity on a beam, AMG preconditioner with 2 x block Gauss-Seidel on the coarse levels and
1 x Gauss-Seidel on level 0, PCG to 1e-12.  Needs a GPU (the apply path has no CPU fallback).

    python examples/elasticity_simple.py [nx ny nz] [--edge-mats] [--robust] [--improve K]

--edge-mats: the energy's edge matrices + matrix-valued smoothed prolongation (what the reference's elasticity preconditioner
builds; ngs_amg_edge_mats), --robust: energy-based strength of connection on top (ngs_amg_crs_robust), --improve K: K smoothing
steps on the prolongation inside its graph (ngs_amg_sp_improve_its).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ngsamg_amd import NgsAMG, fem, Matrix          # noqa: E402
from ngsamg_amd.harness import Solve                 # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    shape = tuple(int(a) for a in argv[:3]) if len(argv) >= 3 else (41, 9, 9)
    # E = 1e3, nu = 0.15 as in the reference example  ->  Lame parameters
    E, nu = 1e3, 0.15
    mu, lam = E / (2 * (1 + nu)), E * nu / ((1 + nu) * (1 - 2 * nu))
    p = fem.elasticity_fast(shape, dirichlet="left", mu=mu, lam=lam, extent=(10.0, 2.0, 2.0))
    a = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    pc_opts = {
        "ngs_amg_max_levels": 30,
        "ngs_amg_max_coarse_size": 10,
        "ngs_amg_sp_omega": 0.8,
        "ngs_amg_sp_max_per_row": 5,
        "ngs_amg_sm_type": "bgs",            # 2 x block Gauss-Seidel over the aggregates ...
        "ngs_amg_sm_steps": 2,
        "ngs_amg_sm_type_spec": ["gs"],      # ... but 1 x Gauss-Seidel on level 0
        "ngs_amg_sm_steps_spec": [1],
        "ngs_amg_log_level": "basic",
        "ngs_amg_do_test": True,
    }
    if "--edge-mats" in sys.argv or "--robust" in sys.argv:
        pc_opts["ngs_amg_edge_mats"] = True
    if "--robust" in sys.argv:
        pc_opts["ngs_amg_crs_robust"] = True
    if "--improve" in sys.argv:
        pc_opts["ngs_amg_sp_improve_its"] = int(sys.argv[sys.argv.index("--improve") + 1])
    c = NgsAMG.elast_3d(a, p.free, coords=p.coords, **pc_opts)
    print(c.GetHierarchy().summary())
    Solve(c, p.load, ms=100, tol=1e-12)


if __name__ == "__main__":
    main()
