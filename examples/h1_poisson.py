#!/usr/bin/env python3
"""3D Poisson with the H1 AMG preconditioner through the drop-in surface (the reference's tests/h1 cases on this build's
stand-in FEM assembly): python examples/h1_poisson.py [n] [smoother: gs | jacobi | bgs]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ngsamg_amd import ngs_amg, fem, Matrix          # noqa: E402
from ngsamg_amd.harness import Solve                  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    sm = sys.argv[2] if len(sys.argv) > 2 else "gs"
    p = fem.poisson_fast((n, n, n), dirichlet="right|top")
    a = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    c = ngs_amg.Preconditioner(a, "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_sm_type=sm, ngs_amg_max_coarse_size=50,
                               ngs_amg_log_level="basic")
    print(c.GetHierarchy().summary())
    Solve(c, p.load, ms=100, tol=1e-10)


if __name__ == "__main__":
    main()
