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

    @property
    def n(self):
        return self.A.n_rows

    @property
    def bs(self):
        return self.A.br


# flag names follow the reference (prefix ngs_amg_, SURVEY.md section 5 "Config / flags")
_OPTION_KEYS = {
    "max_levels": int, "max_coarse_size": int, "first_aaf": float, "aaf": float, "enable_sp": int,
    "sp_omega": float, "sp_max_per_row": int, "sp_min_frac": float, "soc_thresh": float, "max_rounds": int,
    "regularize_cmats": int, "log_level": int,
}


def make_options(dim, energy, **kw):
    lib = _lib.host()
    o = _lib.amgh_options()
    lib.amgh_default_options(C.byref(o), int(dim), int(energy))
    for k, v in kw.items():
        key = k[len("ngs_amg_"):] if k.startswith("ngs_amg_") else k
        if key in _OPTION_KEYS and v is not None:
            if key == "log_level" and isinstance(v, str):
                v = {"none": 0, "basic": 1, "normal": 2, "extra": 3, "debug": 4}.get(v, 0)
            setattr(o, key, _OPTION_KEYS[key](v))
    return o


class Hierarchy:
    """Frozen multigrid hierarchy on the host."""

    def __init__(self, A: Matrix, free=None, coords=None, dim=3, energy=0, **options):
        lib = _lib.host()
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
