"""ctypes bindings of the two native libraries (include/amgh.h, include/amgx.h).

The libraries are built in-tree by ``__graft_entry__.build()``.  There is deliberately no Python or CPU
fallback for the device library: if ``libngsamg_hip.so`` is missing or fails to load, every use of the
apply path raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.path.join(_HERE, "lib")

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
c_u8p = C.POINTER(C.c_uint8)


class NgsAMGError(RuntimeError):
    """Raised for every non-zero return of the native libraries (the reference throws
    ngcore::Exception, surfaced to Python as RuntimeError)."""


# ---------------------------------------------------------------------------------------------
# host library
# ---------------------------------------------------------------------------------------------

class amgh_matrix(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_cols", C.c_int64), ("br", C.c_int32), ("bc", C.c_int32),
                ("rowptr", c_i64p), ("col", c_i32p), ("val", c_f64p)]


class amgh_options(C.Structure):
    _fields_ = [("max_levels", C.c_int32), ("max_coarse_size", C.c_int64), ("first_aaf", C.c_double),
                ("aaf", C.c_double), ("enable_sp", C.c_int32), ("sp_omega", C.c_double),
                ("sp_max_per_row", C.c_int32), ("sp_min_frac", C.c_double), ("soc_thresh", C.c_double),
                ("max_rounds", C.c_int32), ("regularize_cmats", C.c_int32), ("dim", C.c_int32),
                ("energy", C.c_int32), ("log_level", C.c_int32), ("enable_multistep", C.c_int32), ("robust_soc", C.c_int32),
                ("spw", C.c_int32), ("spw_rounds", C.c_int32), ("spw_orphan_round", C.c_int32),
                ("prol_type", C.c_int32), ("sp_max_per_row_classic", C.c_int32), ("edge_mats", C.c_int32), ("crs_robust", C.c_int32), ("spw_cbs", C.c_int32), ("sp_improve_its", C.c_int32), ("prol_only", C.c_int32), ("spw_pick_robust", C.c_int32), ("spw_neib_boost", C.c_int32), ("spw_pick_avg", C.c_int32), ("spw_diag_stab_boost", C.c_double), ("carry_mesh", C.c_int32)]


class amgh_level(C.Structure):
    _fields_ = [("A", amgh_matrix), ("P", amgh_matrix), ("PT", amgh_matrix), ("free", c_u8p),
                ("dinv", c_f64p), ("coords", c_f64p), ("color", c_i32p), ("n_colors", C.c_int32),
                ("agg", c_i32p)]


_host = None
_hip = None


def _load(name):
    # NGSAMG_HIP_LIB: alternative build of the device library (kernel A/B experiments, tools/ab_cycle.py)
    path = os.path.join(LIBDIR, name)
    if name == "libngsamg_hip.so" and os.environ.get("NGSAMG_HIP_LIB"):
        path = os.environ["NGSAMG_HIP_LIB"]
    if not os.path.exists(path):
        raise NgsAMGError(f"{path} not found - run `python __graft_entry__.py` (build()) first")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def host():
    global _host
    if _host is not None:
        return _host
    lib = _load("libngsamg_host.so")
    vp = C.c_void_p
    lib.amgh_last_error.restype = C.c_char_p
    lib.amgh_default_options.argtypes = [C.POINTER(amgh_options), C.c_int, C.c_int]
    lib.amgh_default_options.restype = None
    lib.amgh_setup.argtypes = [C.POINTER(amgh_matrix), c_u8p, c_f64p, C.POINTER(amgh_options), C.POINTER(vp)]
    lib.amgh_n_levels.argtypes = [vp]
    lib.amgh_level_get.argtypes = [vp, C.c_int, C.POINTER(amgh_level)]
    lib.amgh_coarse_inverse.argtypes = [vp, c_i64p, C.POINTER(c_f64p)]
    lib.amgh_log.argtypes = [vp]
    lib.amgh_log.restype = C.c_char_p
    lib.amgh_destroy.argtypes = [vp]
    lib.amgh_destroy.restype = None
    lib.amgh_calc_dinv.argtypes = [C.POINTER(amgh_matrix), c_u8p, C.c_int, c_f64p]
    lib.amgh_coloring.argtypes = [C.POINTER(amgh_matrix), c_u8p, c_i32p, c_i32p]
    lib.amgh_robust_pair_soc.argtypes = [C.c_int32, c_f64p, c_f64p, c_f64p]
    lib.amgh_coloring_blocked.argtypes = [C.POINTER(amgh_matrix), c_u8p, C.c_int64, c_i32p, c_i32p]
    lib.amgh_hybrid_dinv.argtypes = [C.POINTER(amgh_matrix), c_u8p, C.c_int64, c_f64p]
    lib.amgh_hybrid_dinv_ext.argtypes = [C.POINTER(amgh_matrix), c_u8p, C.c_int64, c_f64p, c_f64p]
    lib.amgh_hybrid_dinv_block.argtypes = [C.POINTER(amgh_matrix), c_u8p, C.c_int64, C.c_int, c_f64p]
    lib.amgh_compact_blocks.argtypes = [C.POINTER(amgh_matrix), c_u8p, C.c_int32, C.c_int32, c_i32p, c_i64p]
    lib.amgh_coloring_blockids.argtypes = [C.POINTER(amgh_matrix), c_u8p, c_i32p, c_i32p, c_i32p]
    lib.amgh_hybrid_dinv_block_ids.argtypes = [C.POINTER(amgh_matrix), c_u8p, c_i32p, C.c_int, c_f64p]
    lib.amgh_bgs_dinv.argtypes = [C.POINTER(amgh_matrix), C.c_int32, c_i32p, c_i32p, C.c_int, c_i64p, c_f64p]
    lib.amgh_bgs_coloring.argtypes = [C.POINTER(amgh_matrix), C.c_int32, c_i32p, c_i32p, c_i32p, c_i32p]
    lib.amgh_transpose_count.argtypes = [C.POINTER(amgh_matrix), c_i64p]
    lib.amgh_transpose_fill.argtypes = [C.POINTER(amgh_matrix), c_i64p, c_i32p, c_f64p]
    lib.amgh_matmul.argtypes = [C.POINTER(amgh_matrix), C.POINTER(amgh_matrix), c_i64p, c_i32p, c_f64p]
    lib.amgh_set_galerkin_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.amgh_kuhn_pattern.argtypes = [C.c_int, c_i64p, c_i64p]
    lib.amgh_kuhn_assemble.argtypes = [C.c_int, c_i64p, c_f64p, C.c_int, C.c_int, C.c_double, C.c_double,
                                       c_f64p, c_i64p, c_i32p, c_f64p, c_f64p]
    _host = lib
    return lib


def hcheck(rc):
    if rc != 0:
        raise NgsAMGError(host().amgh_last_error().decode())


# ---------------------------------------------------------------------------------------------
# numpy <-> ctypes helpers
# ---------------------------------------------------------------------------------------------

def ptr(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def as_array(p, n, dtype):
    """numpy view (no copy) of n items behind a ctypes pointer; None for NULL / n == 0."""
    if not p or n == 0:
        return np.empty(0, dtype=dtype)
    return np.ctypeslib.as_array(p, shape=(int(n),)).view(dtype)


class Matrix:
    """(block-)CSR matrix on the host: int64 rowptr, int32 col, fp64 row-major blocks."""

    def __init__(self, n_rows, n_cols, br, bc, rowptr, col, val, owner=None):
        self.n_rows, self.n_cols, self.br, self.bc = int(n_rows), int(n_cols), int(br), int(bc)
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64).reshape(-1)
        self._owner = owner     # keeps the native hierarchy alive for zero-copy views
        if self.rowptr.shape[0] != self.n_rows + 1:
            raise NgsAMGError("Matrix: rowptr has wrong length")
        if self.col.shape[0] != self.nnz or self.val.shape[0] != self.nnz * self.br * self.bc:
            raise NgsAMGError("Matrix: col/val have wrong length")

    @property
    def nnz(self):
        return int(self.rowptr[-1]) if self.rowptr.size else 0

    @property
    def shape(self):
        return (self.n_rows * self.br, self.n_cols * self.bc)

    def desc(self, cls=amgh_matrix):
        d = cls()
        d.n_rows, d.n_cols, d.br, d.bc = self.n_rows, self.n_cols, self.br, self.bc
        d.rowptr = ptr(self.rowptr, C.c_int64)
        d.col = ptr(self.col, C.c_int32)
        d.val = ptr(self.val, C.c_double)
        return d

    @classmethod
    def from_desc(cls, d, owner):
        if d.n_rows == 0 and not d.rowptr:
            return None
        rowptr = as_array(d.rowptr, d.n_rows + 1, np.int64)
        nnz = int(rowptr[-1])
        return cls(d.n_rows, d.n_cols, d.br, d.bc, rowptr, as_array(d.col, nnz, np.int32),
                   as_array(d.val, nnz * d.br * d.bc, np.float64), owner=owner)

    def to_scipy(self):
        import scipy.sparse as sp
        # copies: the arrays may be zero-copy views into the native hierarchy and must not be edited through scipy
        if self.br == 1 and self.bc == 1:
            return sp.csr_matrix((self.val.copy(), self.col.copy(), self.rowptr.copy()), shape=self.shape)
        return sp.bsr_matrix((self.val.reshape(-1, self.br, self.bc).copy(), self.col.copy(), self.rowptr.copy()),
                             shape=self.shape).tocsr()

    @classmethod
    def from_scipy(cls, A, bs=1):
        import scipy.sparse as sp
        if bs == 1:
            A = sp.csr_matrix(A)
            A.sort_indices()
            return cls(A.shape[0], A.shape[1], 1, 1, A.indptr, A.indices, A.data)
        B = sp.bsr_matrix(A, blocksize=(bs, bs))
        B.sort_indices()
        return cls(B.shape[0] // bs, B.shape[1] // bs, bs, bs, B.indptr, B.indices, B.data)


# ---------------------------------------------------------------------------------------------
# device library (include/amgx.h)
# ---------------------------------------------------------------------------------------------

class amgx_matrix(C.Structure):
    _fields_ = amgh_matrix._fields_


class amgx_level_desc(C.Structure):
    _fields_ = [("A", amgx_matrix), ("P", amgx_matrix), ("PT", amgx_matrix), ("dinv", c_f64p),
                ("free_dofs", c_u8p), ("sm_type", C.c_int32), ("omega", C.c_double), ("sm_steps", C.c_int32),
                ("sm_symm", C.c_int32), ("color", c_i32p), ("n_colors", C.c_int32),
                ("bgs_n_blocks", C.c_int32), ("bgs_block_ptr", c_i32p), ("bgs_block_rows", c_i32p),
                ("bgs_dinv_ptr", c_i64p), ("bgs_dinv", c_f64p), ("bgs_color", c_i32p), ("bgs_n_colors", C.c_int32),
                ("Q", amgx_matrix), ("gs_block_rows", C.c_int32), ("gs_block_ids", c_i32p),
                ("gs_block_color", c_i32p), ("gs_n_block_colors", C.c_int32)]


class amgx_hierarchy_desc(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("levels", C.POINTER(amgx_level_desc)), ("cycle", C.c_int32),
                ("clev", C.c_int32), ("coarse_n", C.c_int64), ("coarse_inv", c_f64p), ("device", C.c_int32),
                ("use_graph", C.c_int32)]


class amgx_halo_desc(C.Structure):
    _fields_ = [("n_peers", C.c_int32), ("peer_rank", c_i32p), ("send_ptr", c_i64p), ("send_idx", c_i32p),
                ("recv_ptr", c_i64p), ("n_interior", C.c_int64)]


class amgx_dist_desc(C.Structure):
    _fields_ = [("top", amgx_hierarchy_desc), ("halo", C.POINTER(amgx_halo_desc)), ("tail", amgx_hierarchy_desc),
                ("counts", c_i64p), ("kmap", c_i64p), ("kmap_len", C.c_int64), ("rank", C.c_int32), ("fold", C.c_int32),
                ("gs_stage", c_i32p)]


class amgx_gss4_desc(C.Structure):
    _fields_ = [("A", amgx_matrix), ("subset", C.POINTER(C.c_uint8)), ("dinv", c_f64p), ("color", c_i32p),
                ("n_colors", C.c_int32), ("device", C.c_int32)]


AMGX_SM_JACOBI, AMGX_SM_GS, AMGX_SM_BGS = 0, 1, 2
AMGX_COMM_RCCL, AMGX_COMM_LOCAL = 0, 1
AMGX_UNIQUE_ID_BYTES = 128
AMGX_CYCLE = {"V": 0, "W": 1, "BS": 2}
AMGX_CLEV_NONE, AMGX_CLEV_INV = 0, 1
AMGX_HOST_PTR, AMGX_DEVICE_PTR, AMGX_NO_GRAPH = 0, 1, 2
AMGX_PCG_SINGLE_REDUCTION = 16

AMGX_SYMBOLS = [
    "amgx_last_error", "amgx_create", "amgx_destroy", "amgx_set_stream", "amgx_synchronize", "amgx_apply",
    "amgx_apply_add", "amgx_smooth", "amgx_smooth_v_from_level", "amgx_jacobi_pre", "amgx_jacobi_post", "amgx_residual",
    "amgx_cycle_down", "amgx_cycle_up",
    "amgx_prolong", "amgx_matvec", "amgx_transfer_f2c",
    "amgx_add_c2f", "amgx_coarse_solve", "amgx_n_levels", "amgx_level_info", "amgx_cycle_info", "amgx_matrix_info",
    "amgx_matrix_stream_bytes", "amgx_time_op", "amgx_pcg", "amgx_gmres",
    "amgx_comm_unique_id", "amgx_comm_create", "amgx_comm_destroy", "amgx_comm_last_error", "amgx_comm_set_stream",
    "amgx_comm_synchronize", "amgx_comm_info", "amgx_comm_graph_info", "amgx_comm_graph_note", "amgx_dist_create", "amgx_dist_destroy", "amgx_dist_apply", "amgx_dist_time_kernel", "amgx_dist_pcg", "amgx_dist_gmres",
    "amgx_dist_rhs_buffer", "amgx_dist_handles", "amgx_halo_create", "amgx_halo_destroy", "amgx_halo_exchange",
    "amgx_gss4_create", "amgx_gss4_destroy", "amgx_gss4_last_error", "amgx_gss4_set_stream", "amgx_gss4_synchronize",
    "amgx_gss4_info", "amgx_gss4_smooth", "amgx_gss4_smooth_res", "amgx_gss4_mult_add",
    "amgx_device_count", "amgx_spgemm", "amgx_galerkin", "amgx_csr_result_fetch",
]

AMGH_SYMBOLS = [
    "amgh_last_error", "amgh_default_options", "amgh_setup", "amgh_n_levels", "amgh_level_get",
    "amgh_coarse_inverse", "amgh_log", "amgh_destroy", "amgh_calc_dinv", "amgh_coloring", "amgh_transpose_count",
    "amgh_transpose_fill", "amgh_matmul", "amgh_kuhn_pattern", "amgh_kuhn_assemble", "amgh_bgs_dinv", "amgh_bgs_coloring",
    "amgh_coloring_blocked", "amgh_hybrid_dinv", "amgh_hybrid_dinv_ext", "amgh_hybrid_dinv_block", "amgh_compact_blocks", "amgh_coloring_blockids", "amgh_hybrid_dinv_block_ids",
    "amgh_set_galerkin_hook", "amgh_robust_pair_soc",
]


def hip():
    """Load libngsamg_hip.so.  Raises if it is missing: the apply path has no fallback."""
    global _hip
    if _hip is not None:
        return _hip
    lib = _load("libngsamg_hip.so")
    vp = C.c_void_p
    dp = C.c_void_p   # vectors travel as raw addresses (host numpy or device torch pointers)
    lib.amgx_last_error.argtypes = [vp]
    lib.amgx_last_error.restype = C.c_char_p
    lib.amgx_create.argtypes = [C.POINTER(amgx_hierarchy_desc), C.POINTER(vp)]
    lib.amgx_destroy.argtypes = [vp]
    lib.amgx_set_stream.argtypes = [vp, vp]
    lib.amgx_synchronize.argtypes = [vp]
    lib.amgx_apply.argtypes = [vp, dp, dp, C.c_int, C.c_int]
    lib.amgx_apply_add.argtypes = [vp, C.c_double, dp, dp, C.c_int]
    lib.amgx_smooth.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.amgx_smooth_v_from_level.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.amgx_jacobi_pre.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    lib.amgx_cycle_down.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    lib.amgx_cycle_up.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    lib.amgx_residual.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    lib.amgx_jacobi_post.argtypes = [vp, C.c_int, dp, dp, dp, C.c_int]
    lib.amgx_prolong.argtypes = [vp, C.c_int, C.c_double, dp, dp, dp, C.c_int]
    lib.amgx_matvec.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    lib.amgx_transfer_f2c.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    lib.amgx_add_c2f.argtypes = [vp, C.c_int, C.c_double, dp, dp, C.c_int]
    lib.amgx_coarse_solve.argtypes = [vp, dp, dp, C.c_int]
    lib.amgx_n_levels.argtypes = [vp]
    lib.amgx_level_info.argtypes = [vp, C.c_int, c_i64p, c_i32p, c_i64p]
    lib.amgx_cycle_info.argtypes = [vp, c_i32p, c_i32p, c_i64p]
    lib.amgx_matrix_info.argtypes = [vp, C.c_int, C.c_int, c_i32p, c_i64p, c_i32p]
    lib.amgx_matrix_stream_bytes.argtypes = [vp, C.c_int, C.c_int, c_i64p]
    lib.amgx_time_op.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_f64p]
    lib.amgx_pcg.argtypes = [vp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, c_f64p, c_i32p]
    lib.amgx_gmres.argtypes = [vp, dp, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_i32p]
    # rank-partitioned hierarchies
    lib.amgx_comm_unique_id.argtypes = [C.c_char_p]
    lib.amgx_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(vp)]
    lib.amgx_comm_destroy.argtypes = [vp]
    lib.amgx_comm_last_error.argtypes = [vp]
    lib.amgx_comm_last_error.restype = C.c_char_p
    lib.amgx_comm_set_stream.argtypes = [vp, vp]
    lib.amgx_comm_synchronize.argtypes = [vp]
    lib.amgx_comm_info.argtypes = [vp, c_i32p, c_i32p, c_i32p, c_i64p]
    lib.amgx_comm_graph_info.argtypes = [vp, c_i32p, c_i64p, c_i64p]
    lib.amgx_comm_graph_note.argtypes = [vp]
    lib.amgx_comm_graph_note.restype = C.c_char_p
    lib.amgx_dist_create.argtypes = [vp, C.POINTER(amgx_dist_desc), C.POINTER(vp)]
    lib.amgx_dist_destroy.argtypes = [vp]
    lib.amgx_dist_apply.argtypes = [vp, C.POINTER(dp), C.POINTER(dp), C.c_int, C.c_int]
    lib.amgx_dist_time_kernel.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_f64p]
    lib.amgx_dist_pcg.argtypes = [vp, C.POINTER(dp), C.POINTER(dp), C.c_double, C.c_int, C.c_int, C.c_int, c_f64p, c_i32p]
    lib.amgx_dist_gmres.argtypes = [vp, C.POINTER(dp), C.POINTER(dp), C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, c_f64p, c_i32p]
    lib.amgx_dist_rhs_buffer.argtypes = [vp, C.POINTER(dp), c_i64p, c_i64p]
    lib.amgx_dist_handles.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.amgx_halo_create.argtypes = [vp, C.POINTER(amgx_halo_desc), C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.amgx_halo_destroy.argtypes = [vp]
    lib.amgx_halo_exchange.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(dp), C.c_int]
    # GSS4
    lib.amgx_gss4_create.argtypes = [C.POINTER(amgx_gss4_desc), C.POINTER(vp)]
    lib.amgx_gss4_destroy.argtypes = [vp]
    lib.amgx_gss4_last_error.argtypes = [vp]
    lib.amgx_gss4_last_error.restype = C.c_char_p
    lib.amgx_gss4_set_stream.argtypes = [vp, vp]
    lib.amgx_gss4_synchronize.argtypes = [vp]
    lib.amgx_gss4_info.argtypes = [vp, c_i64p, c_i64p, c_i64p]
    lib.amgx_gss4_smooth.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    lib.amgx_gss4_smooth_res.argtypes = [vp, C.c_int, dp, dp, C.c_int]
    lib.amgx_gss4_mult_add.argtypes = [vp, C.c_double, dp, dp, C.c_int]
    # setup products on the device
    lib.amgx_device_count.argtypes = [c_i32p]
    lib.amgx_spgemm.argtypes = [C.POINTER(amgx_matrix), C.POINTER(amgx_matrix), C.POINTER(vp), c_i64p, c_i64p]
    lib.amgx_galerkin.argtypes = [C.POINTER(amgx_matrix), C.POINTER(amgx_matrix), C.POINTER(amgx_matrix), C.POINTER(vp), c_i64p, c_i64p]
    lib.amgx_csr_result_fetch.argtypes = [vp, c_i64p, c_i32p, c_f64p]
    _hip = lib
    return lib


_device_setup = None


def device_setup(enable=None, min_rows=None):
    """Hand the Galerkin products of amgh_setup to the device library (amgh_set_galerkin_hook <- amgx_galerkin).

    Called by Hierarchy() before every setup: installs the pair once when a GPU is visible and NGSAMG_DEVICE_SETUP is not 0
    (levels with at least NGSAMG_DEVICE_SETUP_MIN_ROWS fine rows, default 20000).  enable=False removes the hook, enable=True
    insists on it (raises without a GPU).  The device product equals the host product bit for bit (tests/test_gpu_spgemm.py);
    without a GPU the host product runs -- the setup is the cold path, the apply path has no such alternative."""
    global _device_setup
    if enable is None:
        if _device_setup is not None and min_rows is None:
            return _device_setup
        enable = os.environ.get("NGSAMG_DEVICE_SETUP", "1") != "0"
        insist = False
    else:
        insist = bool(enable)
    h = host()
    if not enable:
        hcheck(h.amgh_set_galerkin_hook(None, None, 0))
        _device_setup = False
        return False
    ok = os.path.exists(os.path.join(LIBDIR, "libngsamg_hip.so")) or bool(os.environ.get("NGSAMG_HIP_LIB"))
    n = C.c_int32(0)
    if ok:
        d = hip()
        ok = d.amgx_device_count(C.byref(n)) == 0 and n.value > 0
    if not ok:
        if insist:
            raise NgsAMGError("device_setup: no HIP device (or libngsamg_hip.so missing)")
        _device_setup = False
        return False
    if min_rows is None:
        min_rows = int(os.environ.get("NGSAMG_DEVICE_SETUP_MIN_ROWS", "20000"))
    hcheck(h.amgh_set_galerkin_hook(C.cast(d.amgx_galerkin, C.c_void_p), C.cast(d.amgx_csr_result_fetch, C.c_void_p), int(min_rows)))
    _device_setup = True
    return True
