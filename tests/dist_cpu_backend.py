"""scipy/numpy stage backend for ngsamg_amd.dist.DistributedAMG (TEST INFRASTRUCTURE: lets the rank-partitioned setup,
halo tables and cycle logic run without a GPU, e.g. in world_size-2 gloo tests).  Vectors are CPU torch tensors."""
import numpy as np
import torch

from oracle.pyoracle import Oracle


def _dapply(dinv, v, bs):
    """block-diagonal inverse times vector (AoS blocks of bs entries; dinv: [nblocks * bs * bs] row-major)"""
    if bs == 1:
        return dinv[:v.size] * v
    nb = v.size // bs
    return np.einsum("nij,nj->ni", dinv[:nb * bs * bs].reshape(nb, bs, bs), v.reshape(nb, bs)).reshape(-1)


def cpu_backend(omega=0.9, sm_type="jacobi"):
    tails = {}

    class Ops:
        def __init__(self, top, tail_hier, i):
            self.top = top
            self.A = [L.A.to_scipy() for L in top.levels]
            self.P = [L.P.to_scipy() if L.P is not None else None for L in top.levels]
            self.Q = [L.Q.to_scipy() if getattr(L, "Q", None) is not None else None for L in top.levels]
            if id(tail_hier) not in tails:
                tl, tt = list(tail_hier.levels), {"jacobi": "jacobi", "gs": "gs_mc", "hgs": "gs_mc", "bgs": "bgs_mc"}[sm_type]
                if sm_type == "hgs":
                    from copy import copy
                    tt = []
                    for q, L in enumerate(tl):
                        if getattr(L, "hgs_pre", None) is not None:
                            tl[q] = copy(L)
                            tl[q].dinv = np.ascontiguousarray(L.hgs_dinv[:L.A.n_rows])
                            tt.append("gs_order")
                        else:
                            tt.append("gs_mc")
                tails[id(tail_hier)] = Oracle(tl, sm_type=tt, omega=omega, bgs=[L.bgs for L in tail_hier.levels] if sm_type == "bgs" else None)
            # rank-local smoother objects: one single-level oracle per distributed level (rectangular A, no coarse solve)
            self.loc = [(Oracle([L], sm_type="gs_mc", clev="none") if sm_type == "gs" else
                         Oracle([L], sm_type="gs_order", clev="none") if sm_type == "hgs" else
                         Oracle([L], sm_type="bgs_mc", clev="none", bgs=[L.bgs])) if sm_type in ("gs", "hgs", "bgs") and L.P is not None else None
                        for L in top.levels]
            self.tail = tails[id(tail_hier)]

        def zeros(self, n):
            return torch.zeros(int(n), dtype=torch.float64)

        def index(self, idx):
            return torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64))

        def gather(self, vec, idx, out):
            torch.index_select(vec, 0, idx, out=out)

        def jacobi_pre(self, l, bext, x, r):
            L = self.top.levels[l]
            n = L.A.n_rows * L.A.br
            be = bext.numpy()
            xe = omega * _dapply(L.dinv, be, L.A.br)       # ghost entries of x from ghost dinv * ghost b
            x.numpy()[:] = xe[:n]
            r.numpy()[:] = be[:n] - self.A[l] @ xe

        def cycle_down(self, l, bext, x, bc):
            L = self.top.levels[l]
            n = L.A.n_rows
            be = bext.numpy()
            xe = omega * L.dinv * be
            r = be[:n] - self.A[l] @ xe
            x.numpy()[:] = xe[:n] + omega * L.dinv[:n] * r      # second Jacobi step, no coarse correction yet
            bc.numpy()[:] = self.P[l].T @ r

        def cycle_up(self, l, x, xc_ext):
            x.numpy()[:] += self.Q[l] @ xc_ext.numpy()

        def restrict(self, l, r, bc):
            bc.numpy()[:] = self.P[l].T @ r.numpy()

        def prolong(self, l, x, xc, out):
            out.numpy()[:] = x.numpy() + self.P[l] @ xc.numpy()[: self.P[l].shape[1]]

        def jacobi_post(self, l, text, b, x):
            L = self.top.levels[l]
            n, bs = L.A.n_rows * L.A.br, L.A.br
            te = text.numpy()
            x.numpy()[:] = te[:n] + omega * _dapply(L.dinv, b.numpy() - self.A[l] @ te, bs)

        def tail_apply(self, b, x):
            x.numpy()[:] = self.tail.apply(b.numpy().copy())

        def gs_sweep(self, l, back, xext, b, scratch):
            self.loc[l].smooth(0, xext.numpy(), b.numpy(), scratch.numpy(), False, False, False, bool(back))

        def residual(self, l, xext, b, r):
            r.numpy()[:] = b.numpy() - self.A[l] @ xext.numpy()

    return lambda top, tail_hier, i: Ops(top, tail_hier, i)
