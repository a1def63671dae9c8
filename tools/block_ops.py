#!/usr/bin/env python3
"""per-level down / up step timings of the block (elasticity) V-cycle: python tools/block_ops.py [nv] [rot]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from ngsamg_amd import fem, Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rot = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
    p = fem.elasticity_fast((nv, nv, nv), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
    A = Matrix(p.n, p.n, p.bs, p.bs, p.rowptr, p.col, p.val)
    H = Hierarchy(A, p.free, p.coords, dim=3, energy=1, max_coarse_size=50, regularize_cmats=0 if rot else 1)
    print(H.summary())
    for env in ({}, {"AMGX_NO_BLOCK_FOLD": "1"}):
        os.environ.update(env)
        amg = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
        for k in env:
            del os.environ[k]
        print("variant", env or "folded")
        for l in range(H.n_levels - 1):
            q, a, pm = amg.matrix_info(l, "Q"), amg.matrix_info(l, "A"), amg.matrix_info(l, "P")
            print(f"  level {l}: down {amg.time_op(l, 5, 30) * 1e3:8.1f} us  up {amg.time_op(l, 6, 30) * 1e3:8.1f} us   "
                  f"A {a['fmt']} {a['stream_bytes'] / 1e6:.0f} MB  P {pm['fmt']} {pm['stream_bytes'] / 1e6:.0f} MB  Q {q['fmt']} {q['stream_bytes'] / 1e6:.0f} MB")
        print(f"  cycle {amg.time_op(0, 4, 30) * 1e3:.1f} us")


if __name__ == "__main__":
    main()
