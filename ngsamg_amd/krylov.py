"""Device-resident preconditioned CG: stand-in for ``ngsolve.krylovspace.CGSolver`` as the reference's test
drivers use it (reference tests/h1/amg_utils.py:337-363): ``errors[k] = sqrt(|<C r_k, r_k>|)``, stop at
``errors[k] <= tol * errors[0]``.  Vectors are torch CUDA tensors; the operator and the preconditioner are the
HIP kernels behind include/amgx.h (matvec on level 0, one multigrid cycle)."""
from __future__ import annotations


class CGSolver:
    def __init__(self, mat, pre=None, tol=1e-12, maxsteps=100, callback=None):
        """mat: DeviceAMGMatrix (its level-0 matrix is the operator) or any object with MatVec(0, x, y);
        pre: object with Mult(b, x) or None."""
        self.mat, self.pre, self.tol, self.maxsteps, self.callback = mat, pre, tol, maxsteps, callback
        self.errors = []
        self.iterations = 0

    def Solve(self, rhs, sol=None):
        import torch
        b = rhs
        x = torch.zeros_like(b) if sol is None else sol
        d = torch.empty_like(b)
        w = torch.empty_like(b)
        self.mat.MatVec(0, x, w)
        torch.sub(b, w, out=d)
        if self.pre is not None:
            self.pre.Mult(d, w)
        else:
            w.copy_(d)
        s = w.clone()
        wdn = torch.dot(w, d).item()
        err0 = abs(wdn) ** 0.5
        self.errors = [err0]
        self.iterations = 0
        if err0 == 0.0:
            return x
        for it in range(1, self.maxsteps + 1):
            self.mat.MatVec(0, s, w)
            wd = wdn
            as_s = torch.dot(s, w).item()
            alpha = wd / as_s
            x.add_(s, alpha=alpha)
            d.add_(w, alpha=-alpha)
            if self.pre is not None:
                self.pre.Mult(d, w)
            else:
                w.copy_(d)
            wdn = torch.dot(w, d).item()
            beta = wdn / wd
            s.mul_(beta).add_(w)
            err = abs(wdn) ** 0.5
            self.errors.append(err)
            self.iterations = it
            if self.callback:
                self.callback(it, err)
            if err <= self.tol * err0:
                break
        return x
