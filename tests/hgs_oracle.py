"""Oracle configuration for the block-hybrid Gauss-Seidel smoother (test infrastructure): the serial hybrid GS of
oracle.c (gs_order + gs_block: couplings to other blocks use the sweep-start values) with blocks of B consecutive rows,
block-local colour-major order and the l1-modified diagonal -- exactly the data the device handle was created with."""
from copy import copy

import numpy as np


def hgs_levels(levels, info):
    """levels: hierarchy levels; info: DeviceAMGMatrix.hgs (per level None or dict(B, color, n_colors, dinv)).
    Returns (levels', sm_types) for oracle.pyoracle.Oracle."""
    out, types = [], []
    for lv, h in zip(levels, info):
        if h is None:
            out.append(lv)
            types.append("gs_mc")
            continue
        L = copy(lv)
        n = lv.A.n_rows
        blk = (np.arange(n) // h["B"]).astype(np.int32) if h.get("block_of_row") is None else np.asarray(h["block_of_row"], dtype=np.int32)
        col = np.asarray(h["color"])
        rows = np.nonzero(col >= 0)[0]
        if h.get("block_color") is not None:
            # block-coloured form: exact Gauss-Seidel in the order (block colour, block, in-block colour): no frozen couplings
            bc = np.asarray(h["block_color"], dtype=np.int64)
            nb = int(blk.max()) + 1 if n else 0
            key = (bc[rows] * max(1, nb) + blk[rows]) * (int(h["n_colors"]) + 1) + col[rows]
            L.gs_order = rows[np.argsort(key, kind="stable")].astype(np.int32)
            L.gs_block = None
        else:
            key = blk[rows].astype(np.int64) * (int(h["n_colors"]) + 1) + col[rows]
            L.gs_order = rows[np.argsort(key, kind="stable")].astype(np.int32)
            L.gs_block = blk
        L.dinv = np.ascontiguousarray(h["dinv"][:n * lv.A.br * lv.A.br])
        out.append(L)
        types.append("gs_order")
    return out, types
