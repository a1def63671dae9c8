#!/usr/bin/env python3
"""Same-process A/B of whole-cycle time under different environment toggles of libngsamg_hip (cross-run timings on
gpurun land on different physical GPUs and differ by several per cent).  python tools/ab_cycle.py [nv]"""
import os
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "default": {},
    "no_fold": {"AMGX_NO_FOLD": "1"},         # edit this table for the experiment at hand (DESIGN.md 5.6 lists the switches)
}


def main():
    from ngsamg_amd import fem, Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix
    nv = int(sys.argv[1]) if len(sys.argv) > 1 else 215
    sm = sys.argv[2] if len(sys.argv) > 2 else "jacobi"
    p = fem.poisson_fast((nv, nv, nv))
    H = Hierarchy(Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val), p.free, p.coords, dim=3, energy=0, max_coarse_size=50)
    hs = {}
    for inst in range(int(os.environ.get("AB_INSTANCES", "2"))):                      # two instances per variant: allocation placement alone moves the time by ~2 %
        for name, env in VARIANTS.items():
            if os.environ.get("AB_ONLY") and name != os.environ["AB_ONLY"]:
                continue
            for k, v in env.items():
                os.environ[k] = v
            if sm == "gs" and name == "no_fused_restrict":
                continue
            hs[f"{name}#{inst}"] = DeviceAMGMatrix(H, sm_type=sm, device=0)
            for k in env:
                del os.environ[k]
    res = {k: [] for k in hs}
    for rnd in range(5):
        for name, h in hs.items():
            res[name].append(h.time_op(0, 4, reps=30) * 1e3)
    for name, t in res.items():
        t = sorted(t)
        print(f"{name:22s} median {t[len(t) // 2]:8.1f} us  min {t[0]:8.1f} us  -> {1e6 / t[len(t) // 2]:7.1f} applies/s")


if __name__ == "__main__":
    main()
