import sys, numpy as np
sys.path.insert(0,'/root/repo')
from ngsamg_amd import fem, Matrix, ngs_amg
from ngsamg_amd.harness import Solve
for geom in ("squares","fibers"):
    for jump in (1e1,1e2,1e4,1e6):
        def coef(X):
            x, y = X[..., 0], X[..., 1]
            if geom == "squares":
                inner = ((np.abs(x - 0.3) < 0.1) | (np.abs(x - 0.7) < 0.1)) & ((np.abs(y - 0.3) < 0.1) | (np.abs(y - 0.7) < 0.1))
            else:
                inner = (np.floor(y * 10) % 2 == 1) & (np.abs(x - 0.5) < 0.4)
            return np.where(inner, jump, 1.0)
        diri = "left|right|top|bottom" if geom == "squares" else "top|bottom"
        p = fem.poisson_fast((81, 81), dirichlet=diri, coef=coef)
        a = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
        out=[]
        for sm in ("jacobi","gs","bgs"):
            c = ngs_amg.Preconditioner(a, "ngs_amg.h1_scal", freedofs=p.free, ngs_amg_max_coarse_size=5, ngs_amg_dim=2, ngs_amg_sm_type=sm)
            try:
                sol, cg = Solve(c, p.load, ms=400, tol=1e-6, quiet=True, do_test=False)
                out.append((sm, cg.iterations))
            except AssertionError:
                out.append((sm, ">400"))
        print(geom, jump, out, "levels", c.GetNLevels())
