"""Host AMG hierarchy (cold path): thin Python handle over libngsamg_host (include/amgh.h).

Stands where the reference's ``BaseAMGPC::FinalizeLevel -> BuildAMGMat -> SetUpLevels`` stands
(reference src/base/precond/amg_pc.cpp:420-434, 565-736): consumes the finest (block-)CSR matrix plus
free-dof mask, produces the frozen per-level arrays the apply path uploads once.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import Matrix, NgsAMGError


@dataclass
class Level:
    A: Matrix
    P: Matrix | None
    PT: Matrix | None
    free: np.ndarray       # uint8 [n]
    dinv: np.ndarray       # float64 [n*bs*bs]
    coords: np.ndarray | None
    color: np.ndarray      # int32 [n]
    n_colors: int
    agg: np.ndarray | None
    Q: Matrix | None = None     # rank-partitioned levels only: caller-built (I - w Dinv A) P (amgx_level_desc.Q)
    bgs: "BGSData | None" = None  # block Gauss-Seidel data (Hierarchy.build_bgs)

    @property
    def n(self):
        return self.A.n_rows

    @property
    def bs(self):
        return self.A.br


@dataclass
class BGSData:
    """block Gauss-Seidel data of one level (reference BSmoother, block_gssmoother.cpp:17-150)"""
    n_blocks: int
    block_ptr: np.ndarray      # int32 [n_blocks + 1]
    block_rows: np.ndarray     # int32, block rows grouped by block, ascending inside a block
    dinv_ptr: np.ndarray       # int64 [n_blocks + 1], offsets of the dense inverses
    dinv: np.ndarray           # float64, M_k x M_k column-major per block, M_k = bs * |block k|
    color: np.ndarray          # int32 [n_blocks]
    n_colors: int
    order: np.ndarray | None = None    # explicit visiting order of the blocks (oracle only; default: colour-major)


def bgs_blocks_from_aggregates(agg, free=None):
    """GetGSBlocks (amg_pc_vertex_impl.hpp:1171-1269): block cv = fine vertices k with vmap[k] == cv; vertices
    without a coarse vertex (Dirichlet) are in no block"""
    agg = np.asarray(agg, dtype=np.int64)
    keep = agg >= 0
    if free is not None:
        keep &= np.asarray(free).astype(bool)
    rows = np.nonzero(keep)[0]
    order = np.argsort(agg[rows], kind="stable")
    rows = rows[order]
    nb = int(agg.max()) + 1 if agg.size and agg.max() >= 0 else 0
    ptr = np.zeros(nb + 1, dtype=np.int64)
    np.add.at(ptr, agg[rows] + 1, 1)
    return np.cumsum(ptr).astype(np.int32), rows.astype(np.int32)


def bgs_data(A, block_ptr, block_rows, pinv=False):
    """dense (pseudo-)inverses of the diagonal blocks + colouring of the block graph, through the host library"""
    lib = _lib.host()
    block_ptr = np.ascontiguousarray(block_ptr, dtype=np.int32)
    block_rows = np.ascontiguousarray(block_rows, dtype=np.int32)
    nb = block_ptr.size - 1
    M = np.diff(block_ptr).astype(np.int64) * A.br
    dinv_ptr = np.concatenate([[0], np.cumsum(M * M)]).astype(np.int64)
    dinv = np.zeros(max(1, int(dinv_ptr[-1])), dtype=np.float64)
    color = np.zeros(max(1, nb), dtype=np.int32)
    nc = C.c_int32()
    d = A.desc()
    _lib.hcheck(lib.amgh_bgs_dinv(C.byref(d), nb, _lib.ptr(block_ptr, C.c_int32), _lib.ptr(block_rows, C.c_int32), int(bool(pinv)),
                                  _lib.ptr(dinv_ptr, C.c_int64), _lib.ptr(dinv, C.c_double)))
    _lib.hcheck(lib.amgh_bgs_coloring(C.byref(d), nb, _lib.ptr(block_ptr, C.c_int32), _lib.ptr(block_rows, C.c_int32),
                                      _lib.ptr(color, C.c_int32), C.byref(nc)))
    return BGSData(nb, block_ptr, block_rows, dinv_ptr, dinv, color[:nb], int(nc.value))


# flag names follow the reference (prefix ngs_amg_, SURVEY.md section 5 "Config / flags")
_OPTION_KEYS = {
    "max_levels": int, "max_coarse_size": int, "first_aaf": float, "aaf": float, "enable_sp": int,
    "sp_omega": float, "sp_max_per_row": int, "sp_min_frac": float, "soc_thresh": float, "max_rounds": int,
    "regularize_cmats": int, "log_level": int, "enable_multistep": int, "robust_soc": int,
    "spw": int, "spw_rounds": int, "spw_orphan_round": int, "prol_type": int, "sp_max_per_row_classic": int, "edge_mats": int, "crs_robust": int, "spw_cbs": int, "sp_improve_its": int, "prol_only": int, "spw_pick_robust": int, "spw_neib_boost": int, "spw_pick_avg": int, "spw_diag_stab_boost": float, "carry_mesh": int,
}
_PICK_AVGS = {"min": 0, "geom": 1, "harm": 2, "alg": 3, "max": 4}        # spw_agg.hpp:62-65
_PROL_TYPES = {"piecewise": 0, "aux_smoothed": 1, "semi_aux_smoothed": 2, "own": 3}        # vertex_factory_impl.hpp:123-125
_OPTION_ALIASES = {"spw_orphan_treatment": "spw_orphan_round"}        # the reference's flag name (spw_agg.hpp:60)


def make_options(dim, energy, **kw):
    lib = _lib.host()
    o = _lib.amgh_options()
    lib.amgh_default_options(C.byref(o), int(dim), int(energy))
    for k, v in kw.items():
        key = k[len("ngs_amg_"):] if k.startswith("ngs_amg_") else k
        key = _OPTION_ALIASES.get(key, key)
        if key in _OPTION_KEYS and v is not None:
            if key == "log_level" and isinstance(v, str):
                v = {"none": 0, "basic": 1, "normal": 2, "extra": 3, "debug": 4}.get(v, 0)
            if key == "spw_pick_avg" and isinstance(v, str):
                if v not in _PICK_AVGS:
                    raise NgsAMGError(f"ngs_amg_spw_pick_avg: '{v}' (min | geom | harm | alg | max)")
                v = _PICK_AVGS[v]
            if key == "prol_type" and isinstance(v, str):
                if v not in _PROL_TYPES:
                    raise NgsAMGError(f"ngs_amg_prol_type: '{v}' (piecewise | aux_smoothed | semi_aux_smoothed | own)")
                v = _PROL_TYPES[v]
            setattr(o, key, _OPTION_KEYS[key](v))
    return o


class Hierarchy:
    """Frozen multigrid hierarchy on the host."""

    def __init__(self, A: Matrix, free=None, coords=None, dim=3, energy=0, **options):
        lib = _lib.host()
        _lib.device_setup()          # Galerkin products on the device when one is visible (bit-identical; amgh.h: amgh_set_galerkin_hook)
        self.options = make_options(dim, energy, **options)
        self.dim = int(dim)
        self.energy = int(energy)
        self._handle = C.c_void_p()
        free_a = None if free is None else np.ascontiguousarray(free, dtype=np.uint8)
        if free_a is not None and free_a.shape[0] != A.n_rows:
            raise NgsAMGError("freedofs must have one entry per (block) row")
        coords_a = None if coords is None else np.ascontiguousarray(coords, dtype=np.float64)
        if coords_a is not None and coords_a.size != A.n_rows * dim:
            raise NgsAMGError("coords must be [n, dim]")
        d = A.desc()
        _lib.hcheck(lib.amgh_setup(C.byref(d), _lib.ptr(free_a, C.c_uint8), _lib.ptr(coords_a, C.c_double),
                                   C.byref(self.options), C.byref(self._handle)))
        self.levels: list[Level] = []
        for l in range(lib.amgh_n_levels(self._handle)):
            lv = _lib.amgh_level()
            _lib.hcheck(lib.amgh_level_get(self._handle, l, C.byref(lv)))
            n, bs = lv.A.n_rows, lv.A.br
            self.levels.append(Level(
                A=Matrix.from_desc(lv.A, self), P=Matrix.from_desc(lv.P, self) if lv.P.n_rows else None,
                PT=Matrix.from_desc(lv.PT, self) if lv.PT.n_rows else None,
                free=_lib.as_array(lv.free, n, np.uint8), dinv=_lib.as_array(lv.dinv, n * bs * bs, np.float64),
                coords=_lib.as_array(lv.coords, n * dim, np.float64).reshape(n, dim) if lv.coords else None,
                color=_lib.as_array(lv.color, n, np.int32), n_colors=int(lv.n_colors),
                agg=_lib.as_array(lv.agg, n, np.int32) if lv.agg else None))
        nci = C.c_int64()
        pci = _lib.c_f64p()
        _lib.hcheck(lib.amgh_coarse_inverse(self._handle, C.byref(nci), C.byref(pci)))
        self.coarse_n = int(nci.value)
        self.coarse_inv = _lib.as_array(pci, self.coarse_n * self.coarse_n, np.float64)
        self.log = lib.amgh_log(self._handle).decode()

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            _lib.host().amgh_destroy(h)
            self._handle = None

    @property
    def n_levels(self):
        return len(self.levels)

    def build_bgs(self, pinv=None):
        """attach block Gauss-Seidel data to every smoothed level: blocks = the level's aggregates (GetGSBlocks,
        amg_pc_vertex_impl.hpp:1171-1269), (pseudo-)inverted diagonal blocks, colouring of the block graph.
        pinv: pseudo-inverse like the reference's regularize_cmats (default: the hierarchy's option)"""
        if pinv is None:
            pinv = bool(self.options.regularize_cmats)
        for lv in self.levels[:-1]:
            if lv.bgs is None:
                if lv.agg is None:
                    raise NgsAMGError("block Gauss-Seidel needs the aggregates of the level")
                bp, br = bgs_blocks_from_aggregates(lv.agg, lv.free)
                lv.bgs = bgs_data(lv.A, bp, br, pinv=pinv)
        return [lv.bgs for lv in self.levels]

    def operator_complexity(self):
        """sum_l nnz(A_l)*bs_l^2 / nnz(A_0)*bs_0^2  (reference Logger, base_factory.cpp:83-190)."""
        f = lambda L: L.A.nnz * L.A.br * L.A.bc
        return sum(f(L) for L in self.levels) / f(self.levels[0])

    def summary(self):
        lines = ["AMG Summary", f"  levels: {self.n_levels}, OC = {self.operator_complexity():.3f}"]
        for i, L in enumerate(self.levels):
            lines.append(f"  level {i}: n = {L.n}, bs = {L.bs}, nnz = {L.A.nnz}, nnz/row = {L.A.nnz / max(1, L.n):.1f}"
                         + (f", P nnz/row = {L.P.nnz / max(1, L.n):.2f}" if L.P is not None else "")
                         + f", colors = {L.n_colors}")
        return "\n".join(lines)
