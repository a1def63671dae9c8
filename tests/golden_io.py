"""Load a committed golden fixture (tests/golden/*.npz) back into hierarchy-like objects."""
import os
from types import SimpleNamespace

import numpy as np

from ngsamg_amd._lib import Matrix

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["poisson2d_9", "poisson2d_17", "poisson3d_5", "poisson3d_9", "elast3d_4_bs3", "elast3d_4_bs6",
         "elast3d_5_bs3_edge_mats", "elast3d_4_bs6_edge_mats"]


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    nl = int(z["n_levels"])
    levels = []

    def mat(l, tag):
        key = f"l{l}_{tag}_shape"
        if key not in z:
            return None
        nr, nc, br, bc = (int(v) for v in z[key])
        return Matrix(nr, nc, br, bc, z[f"l{l}_{tag}_rowptr"], z[f"l{l}_{tag}_col"], z[f"l{l}_{tag}_val"])

    for l in range(nl):
        color = z[f"l{l}_color"]
        levels.append(SimpleNamespace(A=mat(l, "A"), P=mat(l, "P"), PT=mat(l, "PT"), free=z[f"l{l}_free"],
                                      dinv=z[f"l{l}_dinv"], color=color, n_colors=int(color.max()) + 1 if color.size else 0,
                                      coords=None, agg=None))
        levels[-1].n = levels[-1].A.n_rows
        levels[-1].bs = levels[-1].A.br
        levels[-1].bgs = None
        if f"l{l}_bgs_block_ptr" in z:
            from ngsamg_amd.hierarchy import BGSData
            col = z[f"l{l}_bgs_color"]
            levels[-1].bgs = BGSData(int(col.size), z[f"l{l}_bgs_block_ptr"], z[f"l{l}_bgs_block_rows"], z[f"l{l}_bgs_dinv_ptr"],
                                     np.ascontiguousarray(z[f"l{l}_bgs_dinv"]) if z[f"l{l}_bgs_dinv"].size else np.zeros(1),
                                     col, int(col.max()) + 1 if col.size else 0)
    return z, levels


class FixtureHierarchy:
    """Duck-types ngsamg_amd.hierarchy.Hierarchy for DeviceAMGMatrix (levels + dense coarse inverse)."""

    def __init__(self, levels):
        self.levels = levels
        L = levels[-1]
        A = L.A.to_scipy().toarray()
        n = A.shape[0]
        f = np.repeat(np.asarray(L.free, dtype=bool), L.A.br)
        inv = np.zeros((n, n))
        if f.any():
            inv[np.ix_(f, f)] = np.linalg.inv(A[np.ix_(f, f)])
        self.coarse_n = n
        self.coarse_inv = np.ascontiguousarray(inv.reshape(-1))

    @property
    def n_levels(self):
        return len(self.levels)
