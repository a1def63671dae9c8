"""Device-resident preconditioned CG: stand-in for ``ngsolve.krylovspace.CGSolver`` as the reference's test
drivers use it (reference tests/h1/amg_utils.py:337-363): ``errors[k] = sqrt(|<C r_k, r_k>|)``, stop at
``errors[k] <= tol * errors[0]``.  Vectors are torch CUDA tensors; the operator and the preconditioner are the
HIP kernels behind include/amgx.h (matvec on level 0, one multigrid cycle)."""
from __future__ import annotations


class CGSolver:
    def __init__(self, mat, pre=None, tol=1e-12, maxsteps=100, callback=None, check_every=1):
        """mat: DeviceAMGMatrix (its level-0 matrix is the operator) or any object with MatVec(0, x, y);
        pre: object with Mult(b, x) or None.
        check_every: the scalars (alpha, beta, error) live on the device; the host looks at the error only every
        `check_every` iterations (1 = the reference's behaviour: test after every iteration)."""
        self.mat, self.pre, self.tol, self.maxsteps, self.callback = mat, pre, tol, maxsteps, callback
        self.check_every = max(1, int(check_every))
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
        wdn = torch.dot(w, d)                        # 0-dim device tensors from here on: no host round trips
        errs = torch.zeros(self.maxsteps + 1, dtype=b.dtype, device=b.device)
        errs[0] = wdn.abs().sqrt()
        err0 = float(errs[0].item())
        self.errors = [err0]
        self.iterations = 0
        if err0 == 0.0:
            return x
        done = 0
        for it in range(1, self.maxsteps + 1):
            self.mat.MatVec(0, s, w)
            wd = wdn
            alpha = wd / torch.dot(s, w)
            x.addcmul_(s, alpha)                     # x += alpha s
            d.addcmul_(w, -alpha)                    # d -= alpha A s
            if self.pre is not None:
                self.pre.Mult(d, w)
            else:
                w.copy_(d)
            wdn = torch.dot(w, d)
            beta = wdn / wd
            s.mul_(beta).add_(w)
            errs[it] = wdn.abs().sqrt()
            self.iterations = it
            if it % self.check_every == 0 or it == self.maxsteps:
                host = errs[done + 1:it + 1].cpu().tolist()
                stop = None
                for k, e in enumerate(host):
                    self.errors.append(e)
                    if self.callback:
                        self.callback(done + 1 + k, e)
                    if e <= self.tol * err0 and stop is None:
                        stop = done + 1 + k
                done = it
                if stop is not None:
                    # iterations beyond the first converged one (only possible with check_every > 1) are harmless extra
                    # work; report the count at which the tolerance was met
                    self.errors = self.errors[:stop + 1]
                    self.iterations = stop
                    break
        return x


class _NativeSolver:
    """common part of the solvers that run completely inside the native library (amgx_pcg / amgx_gmres): hand-written
    BLAS-1 kernels with deterministic reductions, scalars of the recurrences on the device, one scalar read per iteration"""

    def __init__(self, mat, pre=None, tol=1e-12, maxsteps=100):
        from .device import DeviceAMGMatrix
        if not isinstance(mat, DeviceAMGMatrix):
            raise TypeError("native Krylov solvers take a DeviceAMGMatrix (its level-0 matrix is the operator)")
        if pre is not None and getattr(pre, "_dev", pre) is not mat and pre is not mat:
            inner = getattr(getattr(pre, "GetAMGMatrix", lambda: None)(), "_dev", None)
            if inner is not mat:
                raise ValueError("native Krylov solvers precondition with the cycle of the same handle (pre = mat or None)")
        self.mat, self.use_pre, self.tol, self.maxsteps = mat, pre is not None, tol, maxsteps
        self.errors, self.iterations = [], 0

    def _run(self, fn, rhs, sol, extra):
        import ctypes as C
        import numpy as np
        from .device import _Vec, _is_torch
        n = self.mat.sizes[0]
        if sol is None:
            if _is_torch(rhs):
                import torch
                sol = torch.zeros_like(rhs)
            else:
                sol = np.zeros(n)
        vb, vx = _Vec(rhs, n, "rhs"), _Vec(sol, n, "sol", True)
        errs = np.zeros(self.maxsteps + 1)
        it = C.c_int32()
        flags = self.mat._flags(vb, vx) | int(getattr(self, "_extra_flags", 0))
        self.mat._ck(fn(self.mat._h, vb.addr, vx.addr, float(self.tol), int(self.maxsteps), *extra, int(self.use_pre), flags,
                        errs.ctypes.data_as(C.POINTER(C.c_double)), C.byref(it)))
        self.iterations = int(it.value)
        self.errors = errs[: self.iterations + 1].tolist()
        return sol


class NativeCGSolver(_NativeSolver):
    """NGSolve CGSolver stand-in running inside libngsamg_hip (amgx_pcg): err_k = sqrt(|<C r_k, r_k>|), stop at err_k <= tol err_0.
    single_reduction: the Chronopoulos / Gear form of the recurrence (AMGX_PCG_SINGLE_REDUCTION: one reduction point and three
    launches per iteration beside the cycle and the level-0 product instead of five)"""

    def __init__(self, mat, pre=None, tol=1e-12, maxsteps=100, single_reduction=False):
        super().__init__(mat, pre, tol, maxsteps)
        from . import _lib
        self._extra_flags = _lib.AMGX_PCG_SINGLE_REDUCTION if single_reduction else 0

    def Solve(self, rhs, sol=None):
        return self._run(self.mat._lib.amgx_pcg, rhs, sol, ())


class NativeGMResSolver(_NativeSolver):
    """restarted, left-preconditioned GMRES inside libngsamg_hip (amgx_gmres); err_k = |C r_k|"""

    def __init__(self, mat, pre=None, tol=1e-12, maxsteps=100, restart=30):
        super().__init__(mat, pre, tol, maxsteps)
        self.restart = int(restart)

    def Solve(self, rhs, sol=None):
        return self._run(self.mat._lib.amgx_gmres, rhs, sol, (self.restart,))
