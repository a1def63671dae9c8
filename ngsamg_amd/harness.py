"""Driver with the output fields of the reference's test / example harness (reference tests/h1/amg_utils.py:337-363,
examples/elasticity/amg_utils.py:219-235): assemble-free ``Solve`` = Test() + CG on the GPU, printing ``used nits``,
``SOLVE = ... sec`` and ``(scal) dofs / (sec * np)``, and asserting what the reference asserts."""
from __future__ import annotations

import time

import numpy as np


def Solve(pre, rhs, ms=100, tol=1e-6, nocb=True, np_ranks=1, do_test=True, quiet=False):
    """pre: an ngsamg_amd.NgsAMG preconditioner (finalized); rhs: numpy vector.  Returns (solution, cg)."""
    import torch
    from .krylov import CGSolver
    amg = pre.GetAMGMatrix()
    if do_test:
        lam_min, lam_max, kappa = pre.Test()
        if not quiet:
            print(f" lami = {lam_min:.6f}, lama = {lam_max:.6f}, kappa = {kappa:.4f}")
    cb = None if nocb else (lambda k, x: print("it =", k, ", err =", x))
    cg = CGSolver(mat=amg._dev, pre=pre, callback=cb, maxsteps=ms, tol=tol)
    b = torch.from_numpy(np.ascontiguousarray(rhs, dtype=np.float64)).cuda()
    torch.cuda.synchronize()
    ts = time.perf_counter()
    sol = cg.Solve(b)
    torch.cuda.synchronize()
    ts = time.perf_counter() - ts
    n, bs = amg.GetNDof(0)
    if not quiet:
        print("---")
        print("multi-dim ", bs)
        print("(vectorial) ndof ", n)
        print("(scalar) ndof ", bs * n)
        print("used nits = ", cg.iterations)
        print("SOLVE = ", ts, "sec")
        print("(vec) dofs / (sec * np) = ", n / (ts * max(np_ranks, 1)))
        print("(scal) dofs / (sec * np) = ", n * bs / (ts * max(np_ranks, 1)))
        print("---")
    assert cg.errors[-1] < tol * cg.errors[0]
    assert cg.iterations < ms
    return sol.cpu().numpy(), cg
