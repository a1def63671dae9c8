"""ctypes wrapper of the CPU oracle (oracle/oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (ngsamg_amd/) never imports this module.  Status: PARITY UNPINNED (see oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)

SM_JACOBI, SM_GS, SM_BGS = 0, 1, 2
CYCLE_V, CYCLE_W, CYCLE_BS = 0, 1, 2


class orc_matrix(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("br", C.c_int32), ("bc", C.c_int32),
                ("rowptr", c_i64p), ("col", c_i32p), ("val", c_f64p)]


class orc_level(C.Structure):
    _fields_ = [("A", orc_matrix), ("P", orc_matrix), ("PT", orc_matrix), ("free", c_u8p), ("dinv", c_f64p),
                ("sm_type", C.c_int32), ("omega", C.c_double), ("sm_steps", C.c_int32), ("sm_symm", C.c_int32),
                ("gs_order", c_i32p), ("gs_order_len", C.c_int64), ("gs_block", c_i32p),
                ("n_blocks", C.c_int32), ("block_ptr", c_i32p), ("block_rows", c_i32p), ("bdinv_ptr", c_i64p),
                ("bdinv", c_f64p), ("block_order", c_i32p)]


class orc_desc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("levels", C.POINTER(orc_level)), ("cycle", C.c_int32),
                ("clev_inv", C.c_int32)]


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "_build", "liboracle.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing - run __graft_entry__.build() (it compiles oracle/oracle.c)")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.orc_last_error.restype = C.c_char_p
    L.orc_create.argtypes = [C.POINTER(orc_desc), C.POINTER(vp)]
    L.orc_destroy.argtypes = [vp]
    L.orc_destroy.restype = None
    L.orc_apply.argtypes = [vp, c_f64p, c_f64p]
    L.orc_apply_add.argtypes = [vp, C.c_double, c_f64p, c_f64p]
    L.orc_smooth_v_from_level.argtypes = [vp, C.c_int, c_f64p, c_f64p, c_f64p, C.c_int, C.c_int, C.c_int]
    L.orc_smooth.argtypes = [vp, C.c_int, C.c_int, c_f64p, c_f64p, c_f64p, C.c_int, C.c_int, C.c_int]
    L.orc_transfer_f2c.argtypes = [vp, C.c_int, c_f64p, c_f64p]
    L.orc_add_c2f.argtypes = [vp, C.c_int, C.c_double, c_f64p, c_f64p]
    L.orc_coarse_solve.argtypes = [vp, c_f64p, c_f64p]
    L.orc_matvec.argtypes = [vp, C.c_int, c_f64p, c_f64p]
    L.orc_pcg.argtypes = [vp, C.POINTER(orc_matrix), c_f64p, c_f64p, C.c_double, C.c_int, c_f64p, C.POINTER(C.c_int)]
    L.orc_gmres.argtypes = [vp, C.POINTER(orc_matrix), c_f64p, c_f64p, C.c_double, C.c_int, C.c_int, c_f64p, C.POINTER(C.c_int)]
    L.orc_set_threads.argtypes = [C.c_int]
    L.orc_set_threads.restype = None
    L.orc_first_touch.argtypes = [vp]
    L.orc_gss4_create.argtypes = [C.POINTER(orc_matrix), c_u8p, c_f64p, C.POINTER(vp)]
    L.orc_gss4_destroy.argtypes = [vp]
    L.orc_gss4_destroy.restype = None
    L.orc_gss4_rows.argtypes = [vp]
    L.orc_gss4_rows.restype = C.c_int64
    L.orc_gss4_nnz.argtypes = [vp]
    L.orc_gss4_nnz.restype = C.c_int64
    L.orc_gss4_set_order.argtypes = [vp, c_i32p, C.c_int64]
    L.orc_gss4_smooth.argtypes = [vp, C.c_int, c_f64p, c_f64p]
    L.orc_gss4_smooth_res.argtypes = [vp, C.c_int, c_f64p, c_f64p]
    L.orc_gss4_mult_add.argtypes = [vp, C.c_double, c_f64p, c_f64p]
    _LIB = L
    return L


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _mat(m):
    d = orc_matrix()
    if m is None:
        return d
    d.n_rows, d.n_cols, d.br, d.bc = m.n_rows, m.n_cols, m.br, m.bc
    d.rowptr, d.col, d.val = _p(m.rowptr, C.c_int64), _p(m.col, C.c_int32), _p(m.val, C.c_double)
    return d


def color_order(color):
    """colour-major visiting order of the free rows (what the GPU multicolour GS kernel executes)."""
    color = np.asarray(color)
    rows = np.nonzero(color >= 0)[0]
    return np.ascontiguousarray(rows[np.argsort(color[rows], kind="stable")].astype(np.int32))


def _vec(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a


class Oracle:
    """CPU restatement of AMGMatrix over a frozen hierarchy.

    levels: sequence of objects with attributes A, P, PT (matrices with n_rows,n_cols,br,bc,rowptr,col,val),
            free, dinv, color (as ngsamg_amd.hierarchy.Level provides).
    sm_type: 'jacobi' | 'gs' (sequential natural order = reference GSS3) | 'gs_mc' (sequential in the
            colour-major order = same arithmetic as the GPU multicolour kernel) | 'bgs' (block Gauss-Seidel in natural
            block order = reference BSmoother) | 'bgs_mc' (blocks visited colour by colour, like the GPU);
            the block smoothers need bgs = [BGSData or None per level] (ngsamg_amd.hierarchy.bgs_data)
    """

    def __init__(self, levels, sm_type="gs", omega=0.9, sm_steps=1, sm_symm=False, cycle="V", clev="inv",
                 threads=1, bgs=None):
        L = lib()
        self._keep = []
        n = len(levels)
        arr = (orc_level * n)()
        types = sm_type if isinstance(sm_type, (list, tuple)) else [sm_type] * n
        for i, lv in enumerate(levels):
            o = arr[i]
            o.A, o.P, o.PT = _mat(lv.A), _mat(lv.P), _mat(lv.PT)
            free = np.ascontiguousarray(lv.free, dtype=np.uint8)
            dinv = np.ascontiguousarray(lv.dinv, dtype=np.float64)
            self._keep += [free, dinv, lv]
            o.free, o.dinv = _p(free, C.c_uint8), _p(dinv, C.c_double)
            t = types[i]
            o.sm_type = SM_JACOBI if t == "jacobi" else (SM_BGS if t in ("bgs", "bgs_mc") else SM_GS)
            if t in ("bgs", "bgs_mc"):
                g = bgs[i] if bgs is not None else None
                if g is None:
                    if i + 1 < n:
                        raise ValueError("block Gauss-Seidel level without BGSData")
                    o.sm_type = SM_JACOBI          # coarsest level: no smoother is ever called there
                else:
                    bp = np.ascontiguousarray(g.block_ptr, dtype=np.int32)
                    br = np.ascontiguousarray(g.block_rows, dtype=np.int32)
                    dp = np.ascontiguousarray(g.dinv_ptr, dtype=np.int64)
                    dv = np.ascontiguousarray(g.dinv, dtype=np.float64)
                    self._keep += [bp, br, dp, dv]
                    o.n_blocks = int(g.n_blocks)
                    o.block_ptr, o.block_rows = _p(bp, C.c_int32), _p(br, C.c_int32)
                    o.bdinv_ptr, o.bdinv = _p(dp, C.c_int64), _p(dv, C.c_double)
                    blk = getattr(lv, "gs_block", None)           # hybrid (rank-partitioned) block smoother: owner per row
                    if blk is not None:
                        blk = np.ascontiguousarray(blk, dtype=np.int32)
                        self._keep.append(blk)
                        o.gs_block = _p(blk, C.c_int32)
                    if t == "bgs_mc":
                        explicit = getattr(g, "order", None)      # e.g. rank-major, colour-major inside a rank (hybrid)
                        order = np.ascontiguousarray((np.asarray(explicit) if explicit is not None else
                                                      np.argsort(np.asarray(g.color), kind="stable")).astype(np.int32))
                        self._keep.append(order)
                        o.block_order = _p(order, C.c_int32)
            o.omega = omega
            o.sm_steps = int(sm_steps[i] if isinstance(sm_steps, (list, tuple)) else sm_steps)
            o.sm_symm = int(bool(sm_symm[i] if isinstance(sm_symm, (list, tuple)) else sm_symm))
            if t == "gs_mc":
                order = color_order(lv.color)
                self._keep.append(order)
                o.gs_order, o.gs_order_len = _p(order, C.c_int32), order.shape[0]
            if t == "gs_order":          # explicit visiting order (+ optional hybrid blocks) attached to the level
                order = np.ascontiguousarray(lv.gs_order, dtype=np.int32)
                self._keep.append(order)
                o.gs_order, o.gs_order_len = _p(order, C.c_int32), order.shape[0]
                blk = getattr(lv, "gs_block", None)
                if blk is not None:
                    blk = np.ascontiguousarray(blk, dtype=np.int32)
                    self._keep.append(blk)
                    o.gs_block = _p(blk, C.c_int32)
        d = orc_desc()
        d.n_levels = n
        d.levels = arr
        d.cycle = {"V": CYCLE_V, "W": CYCLE_W, "BS": CYCLE_BS}[cycle]
        d.clev_inv = 1 if clev == "inv" else 0
        self._keep.append(arr)
        self._h = C.c_void_p()
        self.levels = levels
        self.sizes = [lv.A.n_rows * lv.A.br for lv in levels]
        self.ext_sizes = [lv.A.n_cols * lv.A.bc for lv in levels]
        L.orc_set_threads(int(threads))
        if L.orc_create(C.byref(d), C.byref(self._h)) != 0:
            raise RuntimeError(L.orc_last_error().decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_destroy(self._h)
            self._h = None

    def first_touch(self):
        """NUMA placement of the level data for the multi-threaded CPU baseline (bench.py)"""
        self._ck(lib().orc_first_touch(self._h))

    @staticmethod
    def set_threads(n):
        lib().orc_set_threads(int(n))

    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())

    def apply(self, b, x=None):
        b = _vec(b)
        if x is None:
            x = np.zeros_like(b)
        self._ck(lib().orc_apply(self._h, _p(b, C.c_double), _p(x, C.c_double)))
        return x

    def apply_add(self, s, b, x):
        b = _vec(b)
        self._ck(lib().orc_apply_add(self._h, float(s), _p(b, C.c_double), _p(x, C.c_double)))
        return x

    def smooth(self, level, x, b, res=None, res_updated=False, update_res=False, x_zero=False, back=False):
        if res is None:
            res = np.zeros_like(x)
        self._ck(lib().orc_smooth(self._h, level, 1 if back else 0, _p(x, C.c_double), _p(_vec(b), C.c_double),
                                  _p(res, C.c_double), int(res_updated), int(update_res), int(x_zero)))
        return x, res

    def smooth_v_from_level(self, level, x, b, res, res_updated, update_res, x_zero):
        self._ck(lib().orc_smooth_v_from_level(self._h, level, _p(x, C.c_double), _p(_vec(b), C.c_double),
                                               _p(res, C.c_double), int(res_updated), int(update_res), int(x_zero)))

    def transfer_f2c(self, level, xf):
        xc = np.zeros(self.sizes[level + 1])
        self._ck(lib().orc_transfer_f2c(self._h, level, _p(_vec(xf), C.c_double), _p(xc, C.c_double)))
        return xc

    def add_c2f(self, level, fac, xf, xc):
        self._ck(lib().orc_add_c2f(self._h, level, float(fac), _p(xf, C.c_double), _p(_vec(xc), C.c_double)))
        return xf

    def coarse_solve(self, rhs):
        x = np.zeros(self.sizes[-1])
        self._ck(lib().orc_coarse_solve(self._h, _p(_vec(rhs), C.c_double), _p(x, C.c_double)))
        return x

    def matvec(self, level, x):
        y = np.zeros(self.sizes[level])
        self._ck(lib().orc_matvec(self._h, level, _p(_vec(x), C.c_double), _p(y, C.c_double)))
        return y

    def pcg(self, b, x0=None, tol=1e-8, maxit=200, precond=True):
        b = _vec(b)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
        errs = np.zeros(maxit + 1)
        it = C.c_int()
        self._ck(lib().orc_pcg(self._h if precond else None, C.byref(_mat(self.levels[0].A)), _p(b, C.c_double),
                               _p(x, C.c_double), float(tol), int(maxit), _p(errs, C.c_double), C.byref(it)))
        return x, it.value, errs[: it.value + 1]


    def gmres(self, b, x0=None, tol=1e-8, maxit=200, restart=30, precond=True):
        """restarted GMRES(restart), left-preconditioned with this hierarchy's cycle (oracle.c orc_gmres, modified Gram-Schmidt)"""
        b = _vec(b)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
        errs = np.zeros(maxit + 1)
        it = C.c_int()
        self._ck(lib().orc_gmres(self._h if precond else None, C.byref(_mat(self.levels[0].A)), _p(b, C.c_double),
                                 _p(x, C.c_double), float(tol), int(maxit), int(restart), _p(errs, C.c_double), C.byref(it)))
        return x, it.value, errs[: it.value + 1]


class OracleGSS4:
    """GSS4 restated on the CPU (oracle/gss4.c; reference gssmoother.cpp:407-583).  A: matrix object with n_rows, n_cols, br, bc,
    rowptr, col, val; subset: [n] mask or None; dinv: [n*bs*bs] inverted (replacement) diagonal blocks.
    order: visiting order of the compressed rows (e.g. colour-major, what the GPU does); None = ascending = the reference."""

    def __init__(self, A, subset, dinv, order=None):
        self._L = lib()
        self._A = A
        self._md = _mat(A)
        self._sub = None if subset is None else np.ascontiguousarray(np.asarray(subset).astype(np.uint8))
        self._dinv = np.ascontiguousarray(np.asarray(dinv, dtype=np.float64).ravel())
        h = C.c_void_p()
        if self._L.orc_gss4_create(C.byref(self._md), _p(self._sub, C.c_uint8), _p(self._dinv, C.c_double), C.byref(h)):
            raise RuntimeError("orc_gss4_create failed")
        self._h = h
        if order is not None:
            o = np.ascontiguousarray(np.asarray(order, dtype=np.int32))
            if self._L.orc_gss4_set_order(self._h, _p(o, C.c_int32), o.size):
                raise RuntimeError("orc_gss4_set_order: the order must list every compressed row once")

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_gss4_destroy(self._h)
            self._h = None

    @property
    def rows(self):
        return int(self._L.orc_gss4_rows(self._h))

    @property
    def nnz(self):
        return int(self._L.orc_gss4_nnz(self._h))

    def smooth(self, x, b, back=False):
        self._L.orc_gss4_smooth(self._h, int(back), _p(x, C.c_double), _p(np.ascontiguousarray(b), C.c_double))

    def smooth_res(self, x, res, back=False):
        self._L.orc_gss4_smooth_res(self._h, int(back), _p(x, C.c_double), _p(res, C.c_double))

    def mult_add(self, s, b, x):
        self._L.orc_gss4_mult_add(self._h, float(s), _p(np.ascontiguousarray(b), C.c_double), _p(x, C.c_double))
