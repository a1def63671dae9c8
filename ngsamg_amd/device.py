"""Device-resident AMG matrix: Python handle over the C ABI of include/amgx.h.

Mirrors the reference's ``AMGMatrix`` (reference src/base/solve/amg_matrix.hpp:14-87, pybind surface
src/base/solve/python_solve.cpp:55-109): ``Mult`` / ``MultAdd`` run one multigrid cycle, ``GetSmoother(level)``
exposes ``Smooth`` / ``SmoothBack`` with the reference's flag contract, ``GetMap()`` the grid transfers.

Vectors may be numpy arrays (host: one H2D and one D2H copy per call, like a CPU caller of the reference)
or torch CUDA tensors (device resident: no copies; the work is enqueued on torch's current stream).
There is no CPU fallback: without the HIP library or without a GPU, construction raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Matrix, NgsAMGError

# "hgs" = Gauss-Seidel in the block-hybrid form (one launch per sweep, amgx_level_desc.gs_block_rows); "gs" = multicolour
_SM = {"jacobi": _lib.AMGX_SM_JACOBI, "gs": _lib.AMGX_SM_GS, "hgs": _lib.AMGX_SM_GS, "bgs": _lib.AMGX_SM_BGS}


def gs_block_rows(A):
    """rows per block of the block-hybrid Gauss-Seidel kernel for a scalar level matrix.  G lanes share a row, each holds
    at most 16 entries (+1 when G = 1) in registers; the workgroup has B * G lanes.  Measured at cfg 2:
    one-lane rows run fastest in 256-row blocks (more workgroups per CU hide the colour phases of their neighbours); rows
    that need several lanes get 512-lane workgroups (1024 until round 3).
    0: the level keeps the multicolour form (rows too long, or a level small enough for the single-workgroup tail)."""
    import os
    if A.br == A.bc and A.br in (2, 3, 6) and A.n_rows == A.n_cols:
        # square-block levels (bgsb_sweep_kernel): a workgroup owns ~128 block rows, a whole number of BSELL slices
        # (64 // bs block rows each); levels with fewer than four such blocks keep the multicolour form
        # (levels below 50 k block rows are latency-bound: a workgroup walks its colour phases one after the other, and half-size
        #  blocks have fewer in-block colours -- A/B at cfg 5, 8 k-row level: 110 + 158 us -> 90 + 101 us per cycle)
        rb = 64 // A.br
        B = int(os.environ.get("AMGX_BGSB_ROWS", "0")) or rb * max(1, (128 if A.n_rows >= 50000 else 64) // rb)
        return B if (A.n_rows >= 4 * B and not os.environ.get("AMGX_NO_BGSB")) else 0
    if A.br != 1 or A.bc != 1 or A.n_rows <= 256:
        return 0
    mx = int(np.diff(A.rowptr).max()) if A.n_rows else 0
    for G in (1, 2, 4, 8, 16):
        if mx <= 16 * G + (1 if G == 1 else 0):
            # (several lanes per row: 512-lane workgroups since round 4 -- on the long-row coarse levels of a reference-shaped hierarchy
            #  one 1024-lane workgroup fills a CU's registers and nothing hides its colour phases: level 1 of cfg 2, backward sweep
            #  299 -> 240 us; PCG iterations at 100^3 with 128 / 64 / 32-row blocks on those levels: 17 / 17 / 17, sequential order 15)
            threads = int(os.environ.get("AMGX_GSB_THREADS", "256") if G == 1 else os.environ.get("AMGX_GSB_THREADS_MULTI", "512"))
            threads = threads if threads in (256, 512, 1024) else 1024
            return max(16, threads // G)
    return 0


def hybrid_gs_data(A, free, B, pinv=False):
    """blocked colouring + inverse of the l1-modified (block) diagonal for blocks of B consecutive rows (host library);
    pinv: block levels whose diagonal blocks are pseudo-inverted (ngs_amg_regularize_cmats)"""
    lib = _lib.host()
    d = A.desc()
    fr = None if free is None else np.ascontiguousarray(free, dtype=np.uint8)
    color = np.full(A.n_rows, -1, dtype=np.int32)
    nc = C.c_int32()
    _lib.hcheck(lib.amgh_coloring_blocked(C.byref(d), _lib.ptr(fr, C.c_uint8), int(B), _lib.ptr(color, C.c_int32), C.byref(nc)))
    if A.br > 1:
        dinv = np.zeros(A.n_rows * A.br * A.br, dtype=np.float64)
        _lib.hcheck(lib.amgh_hybrid_dinv_block(C.byref(d), _lib.ptr(fr, C.c_uint8), int(B), int(bool(pinv)), _lib.ptr(dinv, C.c_double)))
        return color, int(nc.value), dinv
    dinv = np.zeros(A.n_cols, dtype=np.float64)          # (rank-partitioned levels: entries of the ghost columns stay 0)
    _lib.hcheck(lib.amgh_hybrid_dinv(C.byref(d), _lib.ptr(fr, C.c_uint8), int(B), _lib.ptr(dinv, C.c_double)))
    return color, int(nc.value), dinv


def block_colored_gs_data(A, free, B, dinv_plain):
    """square-block levels, block-COLOURED Gauss-Seidel (amgx_level_desc.gs_block_color): sweep blocks = runs of B consecutive block
    rows; in-block colouring as for the hybrid form (amgh_coloring_blocked), a colouring of the BLOCK graph (amgh_bgs_coloring: coupled
    blocks differ) and the plain (pseudo-)inverse of the diagonal blocks -- the sweep is exact Gauss-Seidel in the order (block colour,
    block, in-block colour), so no l1 modification.  Returns (color, n_colors, block_color_of_row, n_block_colors, dinv)."""
    lib = _lib.host()
    d = A.desc()
    fr = None if free is None else np.ascontiguousarray(free, dtype=np.uint8)
    n = A.n_rows
    color = np.full(n, -1, dtype=np.int32)
    nc = C.c_int32()
    _lib.hcheck(lib.amgh_coloring_blocked(C.byref(d), _lib.ptr(fr, C.c_uint8), int(B), _lib.ptr(color, C.c_int32), C.byref(nc)))
    nb = (n + B - 1) // B
    bptr = np.minimum(np.arange(nb + 1, dtype=np.int64) * B, n).astype(np.int32)
    brows = np.arange(n, dtype=np.int32)
    bcol = np.zeros(max(1, nb), dtype=np.int32)
    nbc = C.c_int32()
    _lib.hcheck(lib.amgh_bgs_coloring(C.byref(d), int(nb), _lib.ptr(bptr, C.c_int32), _lib.ptr(brows, C.c_int32), _lib.ptr(bcol, C.c_int32), C.byref(nbc)))
    row_bcol = np.ascontiguousarray(np.repeat(bcol[:nb], B)[:n].astype(np.int32))
    return color, int(nc.value), row_bcol, int(nbc.value), np.ascontiguousarray(dinv_plain, dtype=np.float64)


def hybrid_gs_data_compact(A, free, B, pinv=False):
    """square-block levels: compact sweep blocks of at most B block rows grown over the matrix graph (amgh_compact_blocks)
    instead of runs of consecutive rows, with their colouring and l1-modified block diagonal.
    Returns (block_of_row, color, n_colors, dinv)."""
    lib = _lib.host()
    d = A.desc()
    fr = np.ascontiguousarray(np.ones(A.n_rows, dtype=np.uint8) if free is None else free, dtype=np.uint8)
    blk = np.zeros(A.n_rows, dtype=np.int32)
    nb = C.c_int64()
    _lib.hcheck(lib.amgh_compact_blocks(C.byref(d), _lib.ptr(fr, C.c_uint8), max(1, (7 * int(B)) // 8), int(B), _lib.ptr(blk, C.c_int32), C.byref(nb)))
    color = np.full(A.n_rows, -1, dtype=np.int32)
    nc = C.c_int32()
    _lib.hcheck(lib.amgh_coloring_blockids(C.byref(d), _lib.ptr(fr, C.c_uint8), _lib.ptr(blk, C.c_int32), _lib.ptr(color, C.c_int32), C.byref(nc)))
    dinv = np.zeros(A.n_rows * A.br * A.br, dtype=np.float64)
    _lib.hcheck(lib.amgh_hybrid_dinv_block_ids(C.byref(d), _lib.ptr(fr, C.c_uint8), _lib.ptr(blk, C.c_int32), int(bool(pinv)), _lib.ptr(dinv, C.c_double)))
    return blk, color, int(nc.value), dinv


def _is_torch(v):
    return hasattr(v, "data_ptr") and hasattr(v, "is_cuda")


class _Vec:
    """Resolve a vector argument into (address, flags) and keep it alive for the call."""

    def __init__(self, v, n, name, writable=False):
        self.torch = _is_torch(v)
        if self.torch:
            import torch
            if not v.is_cuda:
                raise NgsAMGError(f"{name}: torch tensors must live on the GPU")
            if v.dtype != torch.float64 or not v.is_contiguous() or v.numel() != n:
                raise NgsAMGError(f"{name}: need a contiguous float64 tensor with {n} entries")
            self.obj = v
            self.addr = v.data_ptr()
        else:
            a = v if isinstance(v, np.ndarray) else np.asarray(v, dtype=np.float64)
            if a.dtype != np.float64 or not a.flags.c_contiguous or a.size != n:
                if writable:
                    raise NgsAMGError(f"{name}: need a contiguous float64 array with {n} entries")
                a = np.ascontiguousarray(a, dtype=np.float64)
                if a.size != n:
                    raise NgsAMGError(f"{name}: need {n} entries, got {a.size}")
            self.obj = a
            self.addr = a.ctypes.data


def hierarchy_desc(hierarchy, sm_type="gs", omega=0.9, sm_steps=1, sm_symm=False, mg_cycle="V", clev="inv", device=0,
                   use_graph=True):
    """amgx_hierarchy_desc over the host arrays of a hierarchy.  Returns (desc, keep): `keep` holds everything the
    descriptor points to and must outlive the amgx_create / amgx_dist_create call."""
    levels = hierarchy.levels
    n = len(levels)
    types = sm_type if isinstance(sm_type, (list, tuple)) else [sm_type] * n
    if len(types) != n:
        raise NgsAMGError("sm_type list must have one entry per level")
    arr = (_lib.amgx_level_desc * n)()
    keep = [arr, hierarchy]
    info = [None] * n            # per level: block-hybrid Gauss-Seidel data actually used (tests configure the oracle with it)
    for i, lv in enumerate(levels):
        d = arr[i]
        d.A = lv.A.desc(_lib.amgx_matrix)
        if lv.P is not None:
            d.P = lv.P.desc(_lib.amgx_matrix)
            d.PT = lv.PT.desc(_lib.amgx_matrix)
        d.dinv = _lib.ptr(lv.dinv, C.c_double)
        d.free_dofs = _lib.ptr(lv.free, C.c_uint8)
        if types[i] not in _SM:
            raise NgsAMGError(f"unknown smoother type '{types[i]}' (jacobi | gs | bgs)")
        d.sm_type = _SM[types[i]]
        d.omega = float(omega)
        d.sm_steps = int(sm_steps[i] if isinstance(sm_steps, (list, tuple)) else sm_steps)      # per level: ..._spec flags
        d.sm_symm = int(bool(sm_symm[i] if isinstance(sm_symm, (list, tuple)) else sm_symm))
        d.color = _lib.ptr(lv.color, C.c_int32)
        d.n_colors = int(lv.n_colors)
        if types[i] == "hgs" and i + 1 < n:
            pre = getattr(lv, "hgs_pre", None)      # rank-partitioned levels: computed by the distributed setup (needs ghost diagonals)
            B = pre["B"] if pre else gs_block_rows(lv.A)
            if B > 0:
                import os
                pinv = bool(getattr(getattr(hierarchy, "options", None), "regularize_cmats", 0))
                blk = None
                bcolr, nbc = None, 0
                # square-block levels big enough to be bandwidth-bound: block-COLOURED sweeps (exact Gauss-Seidel between the sweep
                # blocks as well, one launch per block colour) -- PCG iterations of the sequential order (cfg 3: 16 -> 20 with the
                # hybrid form's frozen couplings between grid lines); small levels keep the single launch of the hybrid form
                bc_min = int(os.environ.get("AMGX_BGSB_BC_MIN_ROWS", "50000"))
                if pre:
                    col, nc, dinv = pre["color"], pre["n_colors"], pre["dinv"]
                elif lv.A.br > 1 and lv.A.n_rows >= bc_min and not os.environ.get("AMGX_BGSB_COMPACT") and not os.environ.get("AMGX_NO_BGSB_BC"):
                    col, nc, bcolr, nbc, dinv = block_colored_gs_data(lv.A, lv.free, B, lv.dinv)
                    # a launch per block colour must still fill the chip (256 CUs): an unstructured coarse level with 2 k sweep blocks
                    # in 9 colours (cfg 3 level 1) would run 230 workgroups per launch -- such levels keep the hybrid form
                    wg_min = int(os.environ.get("AMGX_BGSB_BC_MIN_WG", "1024")) if bc_min > 0 else 0
                    if ((lv.A.n_rows + B - 1) // B) < wg_min * max(1, nbc):
                        bcolr, nbc = None, 0
                        col, nc, dinv = hybrid_gs_data(lv.A, lv.free, B, pinv)
                elif lv.A.br > 1 and os.environ.get("AMGX_BGSB_COMPACT"):
                    # block levels: compact sweep blocks (amgh_compact_blocks) freeze half as many couplings as runs of consecutive
                    # rows (= grid lines) and bring the PCG count to within one of the sequential sweep (cfg 3: 17 vs 16; line
                    # blocks 20), but the in-block colour phases then carry half of A and run one slice per wave and phase:
                    # 250 instead of 467 applications/s at cfg 3, 128 instead of 223 at cfg 5 -- not the default
                    blk, col, nc, dinv = hybrid_gs_data_compact(lv.A, lv.free, B, pinv)
                else:
                    col, nc, dinv = hybrid_gs_data(lv.A, lv.free, B, pinv)
                info[i] = dict(B=B, color=col, n_colors=nc, dinv=dinv, block_of_row=blk, block_color=bcolr, n_block_colors=nbc)
                keep.append(info[i])
                d.color, d.n_colors, d.dinv, d.gs_block_rows = _lib.ptr(col, C.c_int32), nc, _lib.ptr(dinv, C.c_double), B
                if blk is not None:
                    d.gs_block_ids = _lib.ptr(blk, C.c_int32)
                if bcolr is not None:
                    d.gs_block_color, d.gs_n_block_colors = _lib.ptr(bcolr, C.c_int32), nbc
        g = getattr(lv, "bgs", None)
        if types[i] == "bgs" and g is None and i + 1 < n:
            raise NgsAMGError("sm_type 'bgs' needs block data on every smoothed level (Hierarchy.build_bgs())")
        if types[i] == "bgs" and g is not None:
            keep.append(g)
            d.bgs_n_blocks = int(g.n_blocks)
            d.bgs_block_ptr, d.bgs_block_rows = _lib.ptr(g.block_ptr, C.c_int32), _lib.ptr(g.block_rows, C.c_int32)
            d.bgs_dinv_ptr, d.bgs_dinv = _lib.ptr(g.dinv_ptr, C.c_int64), _lib.ptr(g.dinv, C.c_double)
            d.bgs_color, d.bgs_n_colors = _lib.ptr(g.color, C.c_int32), int(g.n_colors)
        q = getattr(lv, "Q", None)        # caller-supplied folded prolongation (rank-partitioned levels, dist.py)
        if q is not None:
            d.Q = q.desc(_lib.amgx_matrix)
    desc = _lib.amgx_hierarchy_desc()
    desc.n_levels = n
    desc.levels = arr
    if mg_cycle not in _lib.AMGX_CYCLE:
        raise NgsAMGError(f"unknown mg_cycle '{mg_cycle}' (V | W | BS)")
    desc.cycle = _lib.AMGX_CYCLE[mg_cycle]
    desc.clev = _lib.AMGX_CLEV_INV if clev == "inv" else _lib.AMGX_CLEV_NONE
    L = levels
    if clev == "inv" and hierarchy.coarse_n == 0 and n > 0 and L[-1].A.n_rows == L[-1].A.n_cols and L[-1].A.n_rows > 0:
        # the host setup hands over a dense inverse for up to 4096 unknowns; a larger coarsest level is inverted by
        # amgx_create on the device (reference: the coarsest matrix is ALWAYS inverted, amg_pc.cpp:843-928)
        desc.coarse_n = L[-1].A.n_rows * L[-1].A.br
        desc.coarse_inv = None
    else:
        desc.coarse_n = hierarchy.coarse_n if clev == "inv" else 0
        desc.coarse_inv = _lib.ptr(hierarchy.coarse_inv, C.c_double) if clev == "inv" else None
    desc.device = int(device)
    desc.use_graph = int(bool(use_graph))
    return desc, keep, info


class DeviceAMGMatrix:
    def __init__(self, hierarchy, sm_type="gs", omega=0.9, sm_steps=1, sm_symm=False, mg_cycle="V",
                 clev="inv", device=0, use_graph=True):
        lib = _lib.hip()
        self._lib = lib
        self._cfg = dict(sm_type=sm_type, omega=omega, sm_steps=sm_steps, sm_symm=sm_symm, mg_cycle=mg_cycle, clev=clev,
                         device=device, use_graph=use_graph)
        self.hierarchy = hierarchy
        desc, self._keep, self.hgs = hierarchy_desc(hierarchy, sm_type, omega, sm_steps, sm_symm, mg_cycle, clev, device, use_graph)
        self._h = C.c_void_p()
        self._owned = True
        if lib.amgx_create(C.byref(desc), C.byref(self._h)) != 0:
            raise NgsAMGError(lib.amgx_last_error(None).decode())
        self._set_sizes()

    def _set_sizes(self):
        levels = self.hierarchy.levels
        self.sizes = [lv.A.n_rows * lv.A.br for lv in levels]
        self.ext_sizes = [lv.A.n_cols * lv.A.bc for lv in levels]      # > sizes on rank-partitioned levels (ghost columns)
        self.n_levels = len(levels)
        self._stream = None

    @classmethod
    def view(cls, handle, hierarchy):
        """borrowed handle (amgx_dist_handles): queries and measurement hooks of a hierarchy owned by a communicator"""
        self = cls.__new__(cls)
        self._lib = _lib.hip()
        self.hierarchy = hierarchy
        self._h = handle
        self._owned = False
        self._keep = []
        self._cfg = {}
        self.hgs = [None] * len(hierarchy.levels)
        self._set_sizes()
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and getattr(self, "_owned", True):
            self._lib.amgx_destroy(h)
        self._h = None

    # ------------------------------------------------------------------------------------------
    def _ck(self, rc):
        if rc != 0:
            raise NgsAMGError(self._lib.amgx_last_error(self._h).decode())

    def _flags(self, *vecs, graph=True):
        t = [v.torch for v in vecs]
        if any(t) and not all(t):
            raise NgsAMGError("mixing host arrays and device tensors in one call is not supported")
        f = _lib.AMGX_DEVICE_PTR if all(t) and t else _lib.AMGX_HOST_PTR
        if all(t) and t:
            import torch
            s = int(torch.cuda.current_stream().cuda_stream)
            if s != self._stream:
                self._ck(self._lib.amgx_set_stream(self._h, C.c_void_p(s)))
                self._stream = s
        if not graph:
            f |= _lib.AMGX_NO_GRAPH
        return f

    def synchronize(self):
        self._ck(self._lib.amgx_synchronize(self._h))

    def _size(self, level):
        if not (0 <= level < self.n_levels):
            raise NgsAMGError(f"level {level} out of range (0..{self.n_levels - 1})")
        return self.sizes[level]

    # AMGMatrix::Mult / MultAdd ---------------------------------------------------------------
    def Mult(self, b, x, graph=True):
        vb, vx = _Vec(b, self.sizes[0], "b"), _Vec(x, self.sizes[0], "x", True)
        self._ck(self._lib.amgx_apply(self._h, vb.addr, vx.addr, 0, self._flags(vb, vx, graph=graph)))
        return x

    def MultAdd(self, s, b, x, graph=True):
        vb, vx = _Vec(b, self.sizes[0], "b"), _Vec(x, self.sizes[0], "x", True)
        self._ck(self._lib.amgx_apply_add(self._h, float(s), vb.addr, vx.addr, self._flags(vb, vx, graph=graph)))
        return x

    MultTrans = Mult
    MultTransAdd = MultAdd

    def apply(self, b):
        """convenience: returns a new vector C b of the same kind as b"""
        if _is_torch(b):
            import torch
            x = torch.empty_like(b)
        else:
            x = np.empty(self.sizes[0])
        return self.Mult(b, x)

    # smoothers ---------------------------------------------------------------------------------
    def Smooth(self, level, x, b, res, res_updated=False, update_res=False, x_zero=False, back=False):
        n = self._size(level)
        vx, vb, vr = _Vec(x, self.ext_sizes[level], "x", True), _Vec(b, n, "b"), _Vec(res, n, "res", True)
        self._ck(self._lib.amgx_smooth(self._h, level, 1 if back else 0, vx.addr, vb.addr, vr.addr,
                                       int(res_updated), int(update_res), int(x_zero), self._flags(vx, vb, vr)))

    def SmoothVFromLevel(self, level, x, b, res, res_updated=False, update_res=False, x_zero=False):
        n = self._size(level)
        vx, vb, vr = _Vec(x, n, "x", True), _Vec(b, n, "b"), _Vec(res, n, "res", True)
        self._ck(self._lib.amgx_smooth_v_from_level(self._h, level, vx.addr, vb.addr, vr.addr, int(res_updated),
                                                    int(update_res), int(x_zero), self._flags(vx, vb, vr)))

    # stage entry points for rank-partitioned levels (ghost entries of the gathered vector filled by the caller) ----
    def JacobiPre(self, level, b_ext, x, r):
        vb, vx, vr = _Vec(b_ext, self.ext_sizes[level], "b"), _Vec(x, self._size(level), "x", True), _Vec(r, self._size(level), "r", True)
        self._ck(self._lib.amgx_jacobi_pre(self._h, level, vb.addr, vx.addr, vr.addr, self._flags(vb, vx, vr)))

    def CycleDown(self, level, b_ext, x, b_coarse):
        """x = two Jacobi steps from zero on b (no coarse correction yet), b_coarse = P^T (b - A w Dinv b)"""
        vb, vx = _Vec(b_ext, self.ext_sizes[level], "b"), _Vec(x, self._size(level), "x", True)
        vc = _Vec(b_coarse, self._size(level + 1), "b_coarse", True)
        self._ck(self._lib.amgx_cycle_down(self._h, level, vb.addr, vx.addr, vc.addr, self._flags(vb, vx, vc)))

    def CycleUp(self, level, x, x_coarse_ext):
        """x += Q x_coarse: coarse-grid correction and Jacobi post-smoothing in one product"""
        vx, vc = _Vec(x, self._size(level), "x", True), _Vec(x_coarse_ext, self.q_cols(level), "x_coarse")
        self._ck(self._lib.amgx_cycle_up(self._h, level, vx.addr, vc.addr, self._flags(vx, vc)))

    def is_folded(self, level):
        return self.matrix_info(level, "Q")["fmt"] is not None

    def q_cols(self, level):
        q = getattr(self.hierarchy.levels[level], "Q", None)
        return q.n_cols if q is not None else self.hierarchy.levels[level].P.n_cols

    def JacobiPost(self, level, x_in_ext, b, x_out):
        vi, vb, vo = _Vec(x_in_ext, self.ext_sizes[level], "x_in"), _Vec(b, self._size(level), "b"), _Vec(x_out, self._size(level), "x_out", True)
        self._ck(self._lib.amgx_jacobi_post(self._h, level, vi.addr, vb.addr, vo.addr, self._flags(vi, vb, vo)))

    def Residual(self, level, x_ext, b, r):
        vx, vb, vr = _Vec(x_ext, self.ext_sizes[level], "x"), _Vec(b, self._size(level), "b"), _Vec(r, self._size(level), "r", True)
        self._ck(self._lib.amgx_residual(self._h, level, vx.addr, vb.addr, vr.addr, self._flags(vx, vb, vr)))

    def Prolong(self, level, fac, x_in, x_coarse, x_out):
        nc = self.hierarchy.levels[level].P.n_cols * self.hierarchy.levels[level].P.bc
        vi, vc, vo = _Vec(x_in, self._size(level), "x_in"), _Vec(x_coarse, nc, "x_coarse"), _Vec(x_out, self._size(level), "x_out", True)
        self._ck(self._lib.amgx_prolong(self._h, level, float(fac), vi.addr, vc.addr, vo.addr, self._flags(vi, vc, vo)))

    # matrices / transfers ---------------------------------------------------------------------
    def MatVec(self, level, x, y):
        n = self._size(level)
        vx, vy = _Vec(x, self.ext_sizes[level], "x"), _Vec(y, n, "y", True)
        self._ck(self._lib.amgx_matvec(self._h, level, vx.addr, vy.addr, self._flags(vx, vy)))
        return y

    def TransferF2C(self, level, x_fine, x_coarse):
        vf, vc = _Vec(x_fine, self._size(level), "x_fine"), _Vec(x_coarse, self._size(level + 1), "x_coarse", True)
        self._ck(self._lib.amgx_transfer_f2c(self._h, level, vf.addr, vc.addr, self._flags(vf, vc)))
        return x_coarse

    def AddC2F(self, level, fac, x_fine, x_coarse):
        vf, vc = _Vec(x_fine, self._size(level), "x_fine", True), _Vec(x_coarse, self._size(level + 1), "x_coarse")
        self._ck(self._lib.amgx_add_c2f(self._h, level, float(fac), vf.addr, vc.addr, self._flags(vf, vc)))
        return x_fine

    def CoarseSolve(self, rhs, x):
        n = self.sizes[-1]
        vr, vx = _Vec(rhs, n, "rhs"), _Vec(x, n, "x", True)
        self._ck(self._lib.amgx_coarse_solve(self._h, vr.addr, vx.addr, self._flags(vr, vx)))
        return x

    # queries ------------------------------------------------------------------------------------
    def GetNLevels(self, rank=0):
        return self._lib.amgx_n_levels(self._h)

    def level_info(self, level):
        n, bs, nnz = C.c_int64(), C.c_int32(), C.c_int64()
        self._ck(self._lib.amgx_level_info(self._h, level, C.byref(n), C.byref(bs), C.byref(nnz)))
        return n.value, bs.value, nnz.value

    def cycle_info(self):
        """how the V-cycle is launched: first level of the single-workgroup tail kernel, first level of the collapsed
        (dense) coarse levels and the size of that dense operator (-1 / 0: not used)"""
        t, d, n = C.c_int32(), C.c_int32(), C.c_int64()
        self._ck(self._lib.amgx_cycle_info(self._h, C.byref(t), C.byref(d), C.byref(n)))
        return {"tail_level": t.value, "dense_level": d.value, "dense_n": n.value}

    def matrix_info(self, level, which):
        fmt, stored, lanes = C.c_int32(), C.c_int64(), C.c_int32()
        self._ck(self._lib.amgx_matrix_info(self._h, level, {"A": 0, "P": 1, "PT": 2, "Apre": 3, "Q": 4, "ApreLW": 5, "QLW": 6}[which], C.byref(fmt),
                                            C.byref(stored), C.byref(lanes)))
        nb = C.c_int64()
        self._ck(self._lib.amgx_matrix_stream_bytes(self._h, level, {"A": 0, "P": 1, "PT": 2, "Apre": 3, "Q": 4, "ApreLW": 5, "QLW": 6}[which], C.byref(nb)))
        return {"fmt": {-1: None, 0: "csrvec", 1: "sell", 2: "bsell", 3: "sellwin", 4: "rigid-body", 5: "sell-lw"}.get(fmt.value, "?"), "stored": stored.value, "lanes": lanes.value,
                "stream_bytes": nb.value}

    def time_op(self, level, op, reps=20):
        ms = C.c_double()
        self._ck(self._lib.amgx_time_op(self._h, level, int(op), int(reps), C.byref(ms)))
        return ms.value


# ---------------------------------------------------------------------------------------------
# byte model of SURVEY.md section 8d / BASELINE.md (algorithmic bytes per cycle)
# ---------------------------------------------------------------------------------------------

def _fetch_result(lib, res, n_rows, n_cols, nnz, br=1, bc=1):
    rp = np.zeros(n_rows + 1, dtype=np.int64)
    col = np.zeros(max(1, nnz), dtype=np.int32)
    val = np.zeros(max(1, nnz) * br * bc)
    if lib.amgx_csr_result_fetch(res, _lib.ptr(rp, C.c_int64), _lib.ptr(col, C.c_int32), _lib.ptr(val, C.c_double)) != 0:
        raise NgsAMGError(lib.amgx_last_error(None).decode())
    return Matrix(n_rows, n_cols, br, bc, rp, col[:nnz], val[:nnz * br * bc])


def device_spmm(A, B):
    """C = A B on the device (amgx_spgemm; (block-)CSR, bit-identical to the host library's amgh_matmul).
    None: the device library does not take this product (result blocks beyond 6 x 6, a row with more than 8192 products)."""
    lib = _lib.hip()
    da, db = A.desc(_lib.amgx_matrix), B.desc(_lib.amgx_matrix)
    res, nr, nnz = C.c_void_p(), C.c_int64(), C.c_int64()
    rc = lib.amgx_spgemm(C.byref(da), C.byref(db), C.byref(res), C.byref(nr), C.byref(nnz))
    if rc == 2:
        return None
    if rc != 0:
        raise NgsAMGError(lib.amgx_last_error(None).decode())
    return _fetch_result(lib, res, int(nr.value), B.n_cols, int(nnz.value), A.br, B.bc)


def device_galerkin(PT, A, P):
    """A_c = (P^T A) P on the device (amgx_galerkin); None as device_spmm."""
    lib = _lib.hip()
    dt, da, dp = PT.desc(_lib.amgx_matrix), A.desc(_lib.amgx_matrix), P.desc(_lib.amgx_matrix)
    res, nr, nnz = C.c_void_p(), C.c_int64(), C.c_int64()
    rc = lib.amgx_galerkin(C.byref(dt), C.byref(da), C.byref(dp), C.byref(res), C.byref(nr), C.byref(nnz))
    if rc == 2:
        return None
    if rc != 0:
        raise NgsAMGError(lib.amgx_last_error(None).decode())
    return _fetch_result(lib, res, int(nr.value), P.n_cols, int(nnz.value), PT.br, P.bc)


def matrix_bytes(M):
    """B(M) = nnz*(8*br*bc + 4) + 4*(n+1): fp64 values, int32 columns, int32 row pointers."""
    if M is None:
        return 0
    return M.nnz * (8 * M.br * M.bc + 4) + 4 * (M.n_rows + 1)


def vcycle_bytes(hierarchy):
    """Algorithmic bytes of one Jacobi V(1,1) cycle:
    per level 2 B(A) + B(P) + B(PT) + 16 b^2 n + 15 V_l + 2 V_{l+1}; coarsest 8 (n b)^2 + 2 V."""
    total = 0
    per_level = []
    L = hierarchy.levels
    for i, lv in enumerate(L):
        V = 8 * lv.bs * lv.n
        if i + 1 < len(L):
            Vc = 8 * L[i + 1].bs * L[i + 1].n
            b = 2 * matrix_bytes(lv.A) + matrix_bytes(lv.P) + matrix_bytes(lv.PT) + 16 * lv.bs * lv.bs * lv.n + 15 * V + 2 * Vc
        else:
            b = 8 * (lv.n * lv.bs) ** 2 + 2 * V
        per_level.append(b)
        total += b
    return total, per_level
