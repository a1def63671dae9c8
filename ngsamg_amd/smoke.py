"""__graft_entry__.smoke(): one small V-cycle application on cuda:0, checked against the CPU oracle."""
from __future__ import annotations

import os
import sys

import numpy as np


def run():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU (cuda:0)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix
    from oracle.pyoracle import Oracle        # checker only

    p = fem.poisson_fast((21, 21, 21))
    A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=20)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(p.n) * p.free
    for sm, osm in (("jacobi", "jacobi"), ("gs", "gs_mc")):
        dev = DeviceAMGMatrix(H, sm_type=sm, device=0)
        ref = Oracle(H.levels, sm_type=osm).apply(b)
        bd = torch.from_numpy(b).cuda()
        xd = torch.empty_like(bd)
        dev.Mult(bd, xd)
        torch.cuda.synchronize()
        x = xd.cpu().numpy()
        rel = np.linalg.norm(x - ref) / np.linalg.norm(ref)
        print(f"[smoke] {sm}: levels={H.n_levels} n={p.n} rel.err vs oracle = {rel:.3e}")
        if not rel < 1e-10:
            raise RuntimeError(f"smoke: GPU V-cycle ({sm}) deviates from the oracle: {rel}")
    print("[smoke] ok")
