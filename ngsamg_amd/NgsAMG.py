"""Python surface of the reference's pybind module ``NgsAMG`` (legacy alias ``ngs_amg``), re-hosted on the
MI355X apply path.

Class / method / kwarg names follow the reference (reference src/base/python/python_amg.hpp:12-105,
src/h1/python_h1.cpp:24-48, src/elasticity/python_elasticity.cpp, src/base/solve/python_solve.cpp:55-109,
src/base/coarsening/python_coarse.cpp:15-121, src/base/smoothers/python_smoothers.cpp:35-387).  NGSolve is
not available here, so the classes work in the reference's "strictly algebraic" mode: they take the
assembled finest matrix (ngsamg_amd.Matrix or scipy.sparse) plus a free-dof mask instead of a BilinearForm,
and vectors are numpy arrays (host) or torch CUDA tensors (device resident).

    pre = NgsAMG.h1_scal(mat, freedofs, ngs_amg_max_coarse_size=5)      # == Preconditioner(mat, "ngs_amg.h1_scal", ...)
    pre.Mult(b, x)                                                      # one V-cycle on the GPU

Flags carry the reference's ``ngs_amg_`` prefix (amg_pc.hpp:168); unknown flags are ignored, like NGSolve's
Flags object does.  Errors surface as RuntimeError (NgsAMGError), like ngcore::Exception does.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from ._lib import Matrix, NgsAMGError
from .hierarchy import Hierarchy
from .device import DeviceAMGMatrix

PREFIX = "ngs_amg_"


def _flags(kwargs):
    out = {}
    for k, v in kwargs.items():
        out[k[len(PREFIX):] if k.startswith(PREFIX) else k] = v
    return out


def _energy_flag(f, energy, name="ngs_amg"):
    """ngs_amg_energy = triv | alg | elmat (amg_pc.cpp:333; reference default alg).  H1: alg is what the setup does (edge weights
    from the matrix entries, BuildAlgMesh_ALG_scal).  Elasticity: an explicit alg selects the reference's edge matrices from the
    assembled matrix and its matrix-valued prolongation (ngs_amg_edge_mats) unless ngs_amg_edge_mats is given;
    elmat needs the element matrices, which do not cross this interface."""
    e = f.get("energy")
    if e is None:
        return {}
    e = str(e).lower()
    if e not in ("triv", "alg", "elmat"):
        raise NgsAMGError(f"{name}: ngs_amg_energy = '{e}' (triv | alg | elmat)")
    if e == "elmat":
        raise NgsAMGError(f"{name}: ngs_amg_energy = 'elmat' needs element matrices (AddElementMatrix), which this interface does not carry")
    if e == "alg" and energy == 1 and "edge_mats" not in f:
        return {"edge_mats": 1}
    return {}


def _as_matrix(mat, bs):
    if isinstance(mat, Matrix):
        return mat
    if hasattr(mat, "tocsr"):
        return Matrix.from_scipy(mat, bs)
    raise NgsAMGError("mat must be an ngsamg_amd.Matrix or a scipy.sparse matrix")


class BaseSmoother:
    """smoothers[level] of an AMGMatrix (reference base_smoother.hpp:43-156): Smooth / SmoothBack with the
    (res_updated, update_res, x_zero) contract, executed by the HIP kernels."""

    def __init__(self, amg, level):
        self._amg, self._level = amg, level

    def _res(self, x, res):
        if res is not None:
            return res
        if hasattr(x, "is_cuda"):
            import torch
            return torch.zeros_like(x)
        return np.zeros_like(x)

    def Smooth(self, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        self._amg._dev.Smooth(self._level, x, rhs, self._res(x, res), res_updated, update_res, x_zero, back=False)

    def SmoothBack(self, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        self._amg._dev.Smooth(self._level, x, rhs, self._res(x, res), res_updated, update_res, x_zero, back=True)

    def SmoothK(self, steps, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        res = self._res(x, res)
        self.Smooth(x, rhs, res, res_updated, update_res, x_zero)          # base_smoother.hpp:87-94
        for _ in range(steps - 1):
            self.Smooth(x, rhs, res, update_res, update_res, False)

    def SmoothBackK(self, steps, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        res = self._res(x, res)
        self.SmoothBack(x, rhs, res, res_updated, update_res, x_zero)
        for _ in range(steps - 1):
            self.SmoothBack(x, rhs, res, update_res, update_res, False)

    def GetMatrix(self):
        return LevelMatrix(self._amg, self._level)

    sys_mat = property(GetMatrix)


class LevelMatrix:
    """GetMatrix(level): y = A_level x on the GPU."""

    def __init__(self, amg, level):
        self._amg, self._level = amg, level
        self.host = amg._hier.levels[level].A

    @property
    def height(self):
        return self.host.n_rows * self.host.br

    width = height

    def Mult(self, x, y):
        return self._amg._dev.MatVec(self._level, x, y)

    def CreateVector(self):
        return np.zeros(self.height)


class BaseDOFMapStep:
    """ProlMap step l -> l+1 (reference dof_map.hpp:252-334)."""

    def __init__(self, amg, level):
        self._amg, self._level = amg, level

    def TransferF2C(self, x_fine, x_coarse):
        return self._amg._dev.TransferF2C(self._level, x_fine, x_coarse)

    def AddC2F(self, fac, x_fine, x_coarse):
        return self._amg._dev.AddC2F(self._level, fac, x_fine, x_coarse)

    def TransferC2F(self, x_fine, x_coarse):
        if hasattr(x_fine, "zero_"):
            x_fine.zero_()
        else:
            x_fine[...] = 0.0
        return self._amg._dev.AddC2F(self._level, 1.0, x_fine, x_coarse)

    F2C = TransferF2C
    C2F = TransferC2F

    def GetProl(self):
        return self._amg._hier.levels[self._level].P

    def Concatenate(self, other):
        """this step followed by `other` as ONE prolongation matrix P_this * P_other (python_coarse.cpp:15-121)"""
        return SparseMM(self.GetProl(), other.GetProl() if hasattr(other, "GetProl") else other)

    def ProjectMatrix(self, mat):
        """Galerkin projection P^T mat P of a fine-level matrix (dof_map.hpp: AssembleMatrix)"""
        P = self.GetProl()
        return SparseMM(self._amg._hier.levels[self._level].PT, SparseMM(_as_matrix(mat, P.br), P))

    def PrintTo(self, stream=None):
        import sys
        P = self.GetProl()
        print(f"ProlMap step {self._level}: {P.n_rows} x {P.n_cols}, blocks {P.br} x {P.bc}, {P.nnz} entries",
              file=stream or sys.stdout)


class DOFMap:
    """Container of the grid-transfer steps (reference dof_map.hpp:87-169)."""

    def __init__(self, amg):
        self._amg = amg

    def GetNLevels(self):
        return self._amg._hier.n_levels

    def GetNSteps(self):
        return self._amg._hier.n_levels - 1

    def GetStep(self, k):
        if not 0 <= k < self.GetNSteps():
            raise NgsAMGError(f"DOFMap has no step {k}")
        return BaseDOFMapStep(self._amg, k)

    def CreateVector(self, level):
        L = self._amg._hier.levels[level]
        return np.zeros(L.n * L.bs)

    def TransferF2C(self, level, x_fine, x_coarse):
        return self.GetStep(level).TransferF2C(x_fine, x_coarse)

    def AddC2F(self, level, fac, x_fine, x_coarse):
        return self.GetStep(level).AddC2F(fac, x_fine, x_coarse)

    def SubMap(self, start):
        """the steps start, start+1, ... as a map of their own (python_coarse.cpp: SubMap)"""
        return self._amg.SubAMGMatrix(start).GetMap()

    def ConcStep(self, la, lb):
        """prolongation from level lb to level la (la < lb) as one sparse matrix: P_la * ... * P_(lb-1)"""
        if not 0 <= la < lb <= self.GetNSteps():
            raise NgsAMGError("ConcStep: need 0 <= la < lb <= number of steps")
        P = self._amg._hier.levels[la].P
        for l in range(la + 1, lb):
            P = SparseMM(P, self._amg._hier.levels[l].P)
        return P

    def TransferAtoB(self, la, lb, vin, vout):
        """move a host vector from level la to level lb through all steps in between (dof_map.cpp:500-573)"""
        cur = np.array(vin, dtype=np.float64)
        if la > lb:
            for l in range(la - 1, lb - 1, -1):
                nxt = self.CreateVector(l)
                self.GetStep(l).TransferC2F(nxt, cur)
                cur = nxt
        else:
            for l in range(la, lb):
                nxt = self.CreateVector(l + 1)
                self.GetStep(l).TransferF2C(cur, nxt)
                cur = nxt
        vout[...] = cur
        return vout


class AMGMatrix:
    """The multigrid cycle as a matrix (reference amg_matrix.hpp:14-87)."""

    def __init__(self, hier, dev):
        self._hier, self._dev = hier, dev

    @property
    def height(self):
        return self._dev.sizes[0]

    width = height

    def Mult(self, b, x):
        return self._dev.Mult(b, x)

    def MultAdd(self, s, b, x):
        return self._dev.MultAdd(s, b, x)

    MultTrans = Mult
    MultTransAdd = MultAdd

    def GetMap(self):
        return DOFMap(self)

    def GetNLevels(self, rank=0):
        return self._hier.n_levels

    def GetNDof(self, level, rank=0):
        L = self._hier.levels[level]
        return L.n, L.bs

    def GetSmoother(self, level=0):
        if not 0 <= level < self._hier.n_levels - 1:
            raise NgsAMGError(f"only have {self._hier.n_levels - 1} smoothers")
        return BaseSmoother(self, level)

    def GetMatrix(self, level=0):
        return LevelMatrix(self, level)

    def SmoothVFromLevel(self, level, x, b, res, res_updated=False, update_res=False, x_zero=False):
        return self._dev.SmoothVFromLevel(level, x, b, res, res_updated, update_res, x_zero)

    def SubAMGMatrix(self, start_level):
        """the cycle that starts at level start_level (python_solve.cpp:55-109): its own device handle over the levels
        start_level .. L-1 with the same smoother / cycle configuration"""
        if not 0 <= start_level < self._hier.n_levels:
            raise NgsAMGError("SubAMGMatrix: level out of range")
        if start_level == 0:
            return self
        sub = _SubHierarchy(self._hier, start_level)
        cfg = dict(self._dev._cfg)
        if isinstance(cfg.get("sm_type"), (list, tuple)):
            cfg["sm_type"] = list(cfg["sm_type"][start_level:])
        return AMGMatrix(sub, DeviceAMGMatrix(sub, **cfg))

    def CINV(self, sol, rhs):
        """restrict rhs to the coarsest level, solve there, prolongate back (amg_matrix.cpp:406-431)"""
        m = self.GetMap()
        L = self._hier.n_levels
        cur = np.array(rhs, dtype=np.float64)
        for l in range(L - 1):
            nxt = m.CreateVector(l + 1)
            m.TransferF2C(l, cur, nxt)
            cur = nxt
        xs = np.zeros_like(cur)
        self._dev.CoarseSolve(cur, xs)
        m.TransferAtoB(L - 1, 0, xs, sol)
        return sol

    def GetBF(self, vec, level=0, dof=0, comp=0, rank=0):
        """basis function (level, dof, comp) prolongated to the finest level (amg_matrix.cpp:434-511)"""
        n, bs = self.GetNDof(level)
        if comp >= bs or dof >= n:
            raise NgsAMGError("GetBF: invalid dof / component")
        e = np.zeros(n * bs)
        e[bs * dof + comp] = 1.0
        return self.GetMap().TransferAtoB(level, 0, e, vec)


class _SubHierarchy:
    """levels start .. L-1 of a hierarchy, as a hierarchy of their own"""

    def __init__(self, hier, start):
        self._parent = hier                     # keeps the level arrays alive
        self.levels = list(hier.levels[start:])
        self.coarse_n, self.coarse_inv = hier.coarse_n, hier.coarse_inv
        self.n_levels = len(self.levels)


class _AMGPreconditioner:
    """Common part of h1_scal / h1_2d / h1_3d / elast_2d / elast_3d (reference BaseAMGPC, amg_pc.hpp:26-228)."""
    _bs = 1
    _dim = 3
    _energy = 0
    _name = "amg"

    def __init__(self, mat=None, freedofs=None, coords=None, device=0, **kwargs):
        self.flags = _flags(kwargs)
        self._freedofs = None
        self._coords = coords
        self._device = device
        self._amg = None
        self.finest_mat = None
        if mat is not None:                       # strict-algebraic constructor (python_h1.cpp:24-33)
            self.InitLevel(freedofs)
            self.FinalizeLevel(mat)

    # ---- lifecycle (amg_pc.cpp:375-434) -----------------------------------------------------------
    def InitLevel(self, freedofs=None):
        self._freedofs = None if freedofs is None else np.ascontiguousarray(np.asarray(freedofs).astype(np.uint8))

    def FinalizeLevel(self, mat=None):
        if mat is None:
            raise NgsAMGError("FinalizeLevel: no matrix given")          # amg_pc.cpp:430
        A = _as_matrix(mat, self._bs)
        if A.br != self._bs and not (self._energy == 1 and A.br in (self._dim, self._dim + self._dim * (self._dim - 1) // 2)):
            raise NgsAMGError(f"{self._name}: matrix block size {A.br} does not fit (expected {self._bs})")
        f = self.flags
        dim = int(f.get("dim", self._dim))
        opts = {k: f[k] for k in ("max_levels", "max_coarse_size", "first_aaf", "aaf", "enable_sp", "sp_omega",
                                  "sp_max_per_row", "sp_min_frac", "soc_thresh", "max_rounds", "log_level", "enable_multistep",
                                  "robust_soc", "spw", "spw_rounds", "spw_orphan_treatment", "prol_type", "sp_max_per_row_classic", "edge_mats", "crs_robust", "spw_cbs", "sp_improve_its", "spw_pick_robust", "spw_neib_boost", "spw_pick_avg", "spw_diag_stab_boost", "carry_mesh") if k in f}
        opts.update(_energy_flag(f, self._energy, self._name))
        if self._energy == 1:
            rots = A.br > dim
            opts["regularize_cmats"] = int(f.get("regularize_cmats", not rots))   # elasticity_pc_impl.hpp:134-139
            if self._coords is None:
                raise NgsAMGError(f"{self._name}: vertex coordinates are needed (coords=...)")
        elif "regularize_cmats" in f:
            opts["regularize_cmats"] = int(f["regularize_cmats"])
        self.finest_mat = A
        hier = Hierarchy(A, self._freedofs, self._coords, dim=dim, energy=self._energy, **opts)
        sm_type = str(f.get("sm_type", "gs")).lower()                     # amg_pc.hpp:63
        spec = f.get("sm_type_spec")
        if sm_type not in ("gs", "jacobi", "bgs"):
            sm_type = "gs"                                                # dyn_block_gs ... fall back to gs (amg_pc_vertex_impl.hpp:587-593)
        types = [sm_type] * hier.n_levels
        if spec:
            for i, t in enumerate(spec[: hier.n_levels]):
                types[i] = t if t in ("gs", "jacobi", "bgs") else "gs"
        # Gauss-Seidel runs in the block-hybrid form by default (one launch per sweep: workgroups sweep blocks of consecutive
        # rows like the ranks of the reference's HybridGSSmoother, gssmoother.cpp:709-861); ngs_amg_gs_hybrid=False selects the
        # multicolour form (exact Gauss-Seidel in colour order, one launch per colour).  Block levels stay multicolour.
        if bool(f.get("gs_hybrid", True)):
            types = ["hgs" if t == "gs" else t for t in types]
        if "bgs" in types:                                               # BuildBGSSmoother(mat, GetGSBlocks(level)), amg_pc.cpp:1060-1072
            hier.build_bgs()
        clev = str(f.get("clev", "inv")).lower()
        # per-level overrides of the first levels, like the reference's SpecOpt flags (amg_pc.cpp:1079-1082 GetOpt(level))
        steps = [int(f.get("sm_steps", 1))] * hier.n_levels
        for i, v in enumerate(list(f.get("sm_steps_spec") or [])[: hier.n_levels]):
            steps[i] = int(v)
        symm = [bool(f.get("sm_symm", False))] * hier.n_levels
        for i, v in enumerate(list(f.get("sm_symm_spec") or [])[: hier.n_levels]):
            symm[i] = bool(v)
        dev = DeviceAMGMatrix(hier, sm_type=types, omega=float(f.get("sm_omega", 0.9)),
                              sm_steps=steps, sm_symm=symm,
                              mg_cycle=str(f.get("mg_cycle", "V")).upper(), clev="inv" if clev == "inv" else "none",
                              device=self._device, use_graph=bool(f.get("use_graph", True)))
        self._amg = AMGMatrix(hier, dev)
        if f.get("do_test", False):
            self.Test()
        return self

    def Update(self):
        pass                                                              # amg_pc_vertex.hpp:74

    def _need(self):
        if self._amg is None:
            raise NgsAMGError("preconditioner not finalized (call FinalizeLevel)")   # amg_pc.cpp:446
        return self._amg

    # ---- BaseMatrix interface (amg_pc.cpp:443-488) ---------------------------------------------------
    def Mult(self, b, x):
        return self._need().Mult(b, x)

    def MultAdd(self, s, b, x):
        return self._need().MultAdd(s, b, x)

    MultTrans = Mult
    MultTransAdd = MultAdd

    @property
    def height(self):
        return self._need().height

    width = height

    def IsComplex(self):
        return False

    def CreateRowVector(self):
        return np.zeros(self.height)

    CreateColVector = CreateRowVector

    # ---- queries (python_amg.hpp:28-101) ------------------------------------------------------------
    def GetNLevels(self, rank=0):
        return self._need().GetNLevels(rank)

    def GetNProcs(self, level=0):
        return 1

    def GetBlockSize(self, level=0):
        return self._need().GetNDof(level)[1]

    def GetNDof(self, level, rank=0):
        return self._need().GetNDof(level)[0]

    def GetNDBS(self, level, rank=0):
        return self._need().GetNDof(level)

    def GetBF(self, vec=None, level=0, dof=0, comp=0, rank=0):
        vec = np.zeros(self.height) if vec is None else vec
        return self._need().GetBF(vec, level, dof, comp, rank)

    def CINV(self, sol=None, rhs=None):
        return self._need().CINV(sol, rhs)

    def GetMap(self):
        return self._need().GetMap()

    def GetSmoother(self, level=0):
        return self._need().GetSmoother(level)

    def GetAMGMatrix(self):
        return self._need()

    def GetHierarchy(self):
        return self._need()._hier

    def Test(self, maxit=60):
        """kappa(C A) estimate by Lanczos on the preconditioned operator, like Preconditioner::Test() /
        ngs_amg_do_test (reference utils_sparseLA.cpp:1317-1360; eigenvalue cut-off 5e-5)."""
        amg = self._need()
        n = amg.height
        free = np.repeat(amg._hier.levels[0].free.astype(bool), amg._hier.levels[0].bs)
        rng = np.random.default_rng(0)
        v = rng.standard_normal(n) * free
        A = amg.GetMatrix(0)
        alphas, betas = [], []
        w, z = np.zeros(n), np.zeros(n)
        # Lanczos for C A in the A-inner product
        A.Mult(v, w)
        v /= np.sqrt(np.dot(v, w))
        v_old = np.zeros(n)
        beta = 0.0
        for _ in range(min(maxit, n)):
            A.Mult(v, w)
            amg.Mult(w * free, z)
            z *= free
            A.Mult(z, w)
            alpha = np.dot(v, w)
            z -= alpha * v + beta * v_old
            A.Mult(z, w)
            b2 = np.dot(z, w)
            alphas.append(alpha)
            if b2 <= 1e-28:
                break
            beta = np.sqrt(b2)
            betas.append(beta)
            v_old, v = v, z / beta
        k = len(alphas)
        T = np.diag(alphas) + np.diag(betas[: k - 1], 1) + np.diag(betas[: k - 1], -1)
        ev = np.linalg.eigvalsh(T)
        ev = ev[ev > 5e-5]
        self.lam_min, self.lam_max = float(ev.min()), float(ev.max())
        self.kappa = self.lam_max / self.lam_min
        return self.lam_min, self.lam_max, self.kappa


class h1_scal(_AMGPreconditioner):
    _bs, _dim, _energy, _name = 1, 3, 0, "h1_scal"


class h1_2d(_AMGPreconditioner):
    _bs, _dim, _energy, _name = 2, 2, 0, "h1_2d"


class h1_3d(_AMGPreconditioner):
    _bs, _dim, _energy, _name = 3, 3, 0, "h1_3d"


class elast_2d(_AMGPreconditioner):
    _bs, _dim, _energy, _name = 3, 2, 1, "elast_2d"


class elast_3d(_AMGPreconditioner):
    _bs, _dim, _energy, _name = 6, 3, 1, "elast_3d"


# ---- registry (reference RegisterPreconditioner<T>("NgsAMG.<name>"), amg_register.hpp:80-98) ------------
_REGISTRY = {}
for _cls in (h1_scal, h1_2d, h1_3d, elast_2d, elast_3d):
    for _prefix in ("NgsAMG.", "ngs_amg."):
        _REGISTRY[_prefix + _cls._name] = _cls


def Preconditioner(mat, name, freedofs=None, **flags):
    """Stand-in for ``ngsolve.Preconditioner(a, "ngs_amg.h1_scal", **flags)`` in strictly algebraic mode."""
    if name not in _REGISTRY:
        raise NgsAMGError(f"unknown preconditioner '{name}' (known: {sorted(_REGISTRY)})")
    return _REGISTRY[name](mat, freedofs, **flags)


# ---- stand-alone smoothers (python_smoothers.cpp:144-387) -------------------------------------------------

class _SingleLevel:
    """hierarchy-like object with one level, so a smoother can live on its own handle"""

    def __init__(self, A, free, pinv):
        import ctypes as C
        lib = _lib.host()
        n, bs = A.n_rows, A.br
        free = np.ones(n, dtype=np.uint8) if free is None else np.ascontiguousarray(np.asarray(free).astype(np.uint8))
        dinv = np.zeros(n * bs * bs)
        d = A.desc()
        _lib.hcheck(lib.amgh_calc_dinv(C.byref(d), _lib.ptr(free, C.c_uint8), int(bool(pinv)), _lib.ptr(dinv, C.c_double)))
        color = np.zeros(n, dtype=np.int32)
        nc = C.c_int32()
        _lib.hcheck(lib.amgh_coloring(C.byref(d), _lib.ptr(free, C.c_uint8), _lib.ptr(color, C.c_int32), C.byref(nc)))
        from .hierarchy import Level
        self.levels = [Level(A=A, P=None, PT=None, free=free, dinv=dinv, coords=None, color=color,
                             n_colors=int(nc.value), agg=None)]
        self.coarse_n = 0
        self.coarse_inv = np.empty(0)
        self.n_levels = 1


class _StandaloneSmoother(BaseSmoother):
    def __init__(self, mat, freedofs, sm_type, pinv=False, nsteps=1, symm=False, omega=0.9, device=0):
        A = _as_matrix(mat, 1)
        hier = _SingleLevel(A, freedofs, pinv)
        dev = DeviceAMGMatrix(hier, sm_type=sm_type, omega=omega, sm_steps=nsteps, sm_symm=symm, clev="none", device=device)
        super().__init__(AMGMatrix(hier, dev), 0)


def CreateJacobiSmoother(mat, freedofs=None, omega=0.9, device=0):
    return _StandaloneSmoother(mat, freedofs, "jacobi", omega=omega, device=device)


def CreateHybridGSS(mat, freedofs=None, pinv=False, NG_MPI_overlap=True, NG_MPI_thread=False, symm=False,
                    symm_loc=False, nsteps=1, nsteps_loc=1, device=0):
    """single GPU: the hybrid smoother degenerates to (multicolour) Gauss-Seidel on the local matrix"""
    return _StandaloneSmoother(mat, freedofs, "gs", pinv=pinv, nsteps=nsteps, symm=symm, device=device)


def CreateHybridBlockGSS(mat, blocks, NG_MPI_overlap=True, NG_MPI_thread=False, shm=True, sl2=False, bs2=True, pinv=False,
                         blocks_no=False, symm=False, symm_loc=False, nsteps=1, nsteps_loc=1, device=0):
    """stand-alone block Gauss-Seidel smoother over caller-given blocks (reference python_smoothers.cpp:197-275;
    single GPU: the hybrid smoother degenerates to BSmoother on the local matrix).  blocks: iterable of row-index lists"""
    from .hierarchy import bgs_data
    A = _as_matrix(mat, 1)
    hier = _SingleLevel(A, None, pinv)
    rows = [np.sort(np.asarray(list(b), dtype=np.int32)) for b in blocks]
    ptr = np.concatenate([[0], np.cumsum([r.size for r in rows])]).astype(np.int32)
    hier.levels[0].bgs = bgs_data(A, ptr, np.concatenate(rows) if rows else np.empty(0, dtype=np.int32), pinv=pinv)
    dev = DeviceAMGMatrix(hier, sm_type="bgs", sm_steps=nsteps, sm_symm=symm, clev="none", device=device)
    return BaseSmoother(AMGMatrix(hier, dev), 0)


class GSS4:
    """Gauss-Seidel on a subset of the rows of a sparse matrix, on a compressed device copy (reference GSS4<TM>,
    gssmoother.hpp:99-143, gssmoother.cpp:407-583; the EX-stage smoother of HybridGSSmoother).

        GSS4(mat, subset=None, repl_diag=None, pinv=False, bs=1)

    repl_diag: replacement diagonal blocks [n, bs, bs] (the hybrid smoother's mod_diag); default = the matrix diagonal.
    Methods as in the reference: Smooth / SmoothBack (x, b), SmoothRES / SmoothBackRES (x, res), MultAdd(s, b, x).
    Vectors: numpy arrays (copied in and out) or CUDA float64 tensors (in place).  Rows of equal colour go in parallel."""

    def __init__(self, mat, subset=None, repl_diag=None, pinv=False, bs=1, device=0):
        import ctypes as C
        from .device import _Vec
        self._Vec = _Vec
        A = _as_matrix(mat, bs)
        self._A = A
        n, b = A.n_rows, A.br
        sub = np.ones(n, dtype=np.uint8) if subset is None else np.ascontiguousarray(np.asarray(subset).astype(np.uint8))
        if sub.size != n:
            raise NgsAMGError("GSS4: subset must have one entry per (block) row")
        host = _lib.host()
        d = A.desc()
        if repl_diag is None:
            dinv = np.zeros(n * b * b)
            _lib.hcheck(host.amgh_calc_dinv(C.byref(d), _lib.ptr(sub, C.c_uint8), int(bool(pinv)), _lib.ptr(dinv, C.c_double)))
        else:
            rd = np.asarray(repl_diag, dtype=np.float64).reshape(n, b, b)
            dinv = np.zeros((n, b, b))
            idx = np.flatnonzero(sub)
            # CalcPseudoInverseTryNormal / CalcInverse on the replacement blocks (gssmoother.cpp:426-434)
            dinv[idx] = np.linalg.pinv(rd[idx], rcond=1e-12, hermitian=False) if pinv else np.linalg.inv(rd[idx])
            dinv = np.ascontiguousarray(dinv.ravel())
        color = np.zeros(n, dtype=np.int32)
        nc = C.c_int32()
        _lib.hcheck(host.amgh_coloring(C.byref(d), _lib.ptr(sub, C.c_uint8), _lib.ptr(color, C.c_int32), C.byref(nc)))
        self.subset, self.dinv, self.color, self.n_colors = sub, dinv, color, int(nc.value)
        self._lib = _lib.hip()
        gd = _lib.amgx_gss4_desc()
        gd.A = A.desc(_lib.amgx_matrix)
        gd.subset = _lib.ptr(sub, C.c_uint8)
        gd.dinv = _lib.ptr(dinv, C.c_double)
        gd.color = _lib.ptr(color, C.c_int32)
        gd.n_colors, gd.device = self.n_colors, int(device)
        h = C.c_void_p()
        if self._lib.amgx_gss4_create(C.byref(gd), C.byref(h)):
            raise NgsAMGError("amgx_gss4_create: " + self._lib.amgx_gss4_last_error(None).decode())
        self._h, self._stream = h, 0
        self._n, self._nc = n * b, A.n_cols * b

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.amgx_gss4_destroy(h)
            self._h = None

    def _ck(self, rc):
        if rc:
            raise NgsAMGError(self._lib.amgx_gss4_last_error(self._h).decode())

    def _flags(self, *vecs):
        import ctypes as C
        t = [v.torch for v in vecs]
        if any(t) and not all(t):
            raise NgsAMGError("mixing host arrays and device tensors in one call is not supported")
        if all(t):
            import torch
            s = int(torch.cuda.current_stream().cuda_stream)
            if s != self._stream:
                self._ck(self._lib.amgx_gss4_set_stream(self._h, C.c_void_p(s)))
                self._stream = s
            return _lib.AMGX_DEVICE_PTR
        return _lib.AMGX_HOST_PTR

    def info(self):
        import ctypes as C
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self._lib.amgx_gss4_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"rows": a.value, "rows_touched": b.value, "nnz": c.value, "colors": self.n_colors}

    def _rhs(self, back, x, b):
        vx, vb = self._Vec(x, self._nc, "x", True), self._Vec(b, self._n, "b")
        self._ck(self._lib.amgx_gss4_smooth(self._h, int(back), vx.addr, vb.addr, self._flags(vx, vb)))

    def _resf(self, back, x, res):
        vx, vr = self._Vec(x, self._n, "x", True), self._Vec(res, self._nc, "res", True)
        self._ck(self._lib.amgx_gss4_smooth_res(self._h, int(back), vx.addr, vr.addr, self._flags(vx, vr)))

    def Smooth(self, x, b):
        self._rhs(False, x, b)

    def SmoothBack(self, x, b):
        self._rhs(True, x, b)

    def SmoothRES(self, x, res):
        self._resf(False, x, res)

    def SmoothBackRES(self, x, res):
        self._resf(True, x, res)

    def MultAdd(self, s, b, x):
        vb, vx = self._Vec(b, self._n, "b"), self._Vec(x, self._n, "x", True)
        self._ck(self._lib.amgx_gss4_mult_add(self._h, float(s), vb.addr, vx.addr, self._flags(vb, vx)))


class _DirectInverseSmoother(BaseSmoother):
    """RichardsonSmoother(A, A^-1 on the free dofs, omega = 1): what CreateHybridDISmoother degenerates to on one
    rank (python_smoothers.cpp:278-312)"""

    def __init__(self, mat, freedofs, device=0):
        A = _as_matrix(mat, 1)
        hier = _SingleLevel(A, freedofs, False)
        n = A.n_rows * A.br
        if n > 4096:
            raise NgsAMGError("CreateHybridDISmoother: the dense inverse is limited to 4096 scalar dofs")
        D = A.to_scipy().toarray()
        f = np.repeat(hier.levels[0].free.astype(bool), A.br)
        inv = np.zeros((n, n))
        inv[np.ix_(f, f)] = np.linalg.inv(D[np.ix_(f, f)])
        hier.coarse_n, hier.coarse_inv = n, np.ascontiguousarray(inv.ravel())
        dev = DeviceAMGMatrix(hier, sm_type="jacobi", clev="inv", device=device)
        super().__init__(AMGMatrix(hier, dev), 0)

    def _step(self, x, rhs, res, res_updated, update_res, x_zero):
        dev = self._amg._dev
        res = self._res(x, res)
        if not res_updated:
            dev.Residual(0, x, rhs, res)
        d = np.zeros_like(np.asarray(res)) if not hasattr(res, "new_zeros") else res.new_zeros(res.shape)
        dev.CoarseSolve(res, d)
        x += d
        if update_res:
            dev.Residual(0, x, rhs, res)

    def Smooth(self, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        self._step(x, rhs, res, res_updated, update_res, x_zero)

    SmoothBack = Smooth


def CreateHybridDISmoother(mat, freedofs=None, NG_MPI_overlap=True, NG_MPI_thread=False, symm=False, nsteps=1, device=0):
    sm = _DirectInverseSmoother(mat, freedofs, device=device)
    return ProxySmoother(sm, nsteps, symm) if (nsteps > 1 or symm) else sm


# ---- utils (reference src/base/utils/python_utils.cpp:30-193) ---------------------------------------------

def SparseMM(A, B):
    """C = A * B for (block-)sparse matrices"""
    import ctypes as C
    A, B = _as_matrix(A, 1), _as_matrix(B, 1)
    if A.n_cols != B.n_rows or A.bc != B.br:
        raise NgsAMGError("SparseMM: shapes do not match")
    lib = _lib.host()
    da, db = A.desc(), B.desc()
    rp = np.zeros(A.n_rows + 1, dtype=np.int64)
    _lib.hcheck(lib.amgh_matmul(C.byref(da), C.byref(db), _lib.ptr(rp, C.c_int64), None, None))
    col = np.zeros(max(1, int(rp[-1])), dtype=np.int32)
    val = np.zeros(max(1, int(rp[-1])) * A.br * B.bc)
    _lib.hcheck(lib.amgh_matmul(C.byref(da), C.byref(db), _lib.ptr(rp, C.c_int64), _lib.ptr(col, C.c_int32), _lib.ptr(val, C.c_double)))
    return Matrix(A.n_rows, B.n_cols, A.br, B.bc, rp, col[: int(rp[-1])], val[: int(rp[-1]) * A.br * B.bc])


def CompressSparseMatrix(mat, tol=1e-20):
    """drop the stored blocks whose entries are all <= tol in magnitude"""
    A = _as_matrix(mat, 1)
    bb = A.br * A.bc
    keep = np.abs(np.asarray(A.val).reshape(-1, bb)).max(axis=1) > tol if A.nnz else np.zeros(0, dtype=bool)
    rows = np.repeat(np.arange(A.n_rows), np.diff(A.rowptr))
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=A.n_rows))]).astype(np.int64)
    return Matrix(A.n_rows, A.n_cols, A.br, A.bc, rp, np.asarray(A.col)[keep].astype(np.int32),
                  np.asarray(A.val).reshape(-1, bb)[keep].ravel())


def ToSparseMatrix(mat, compress=False, compressTol=1e-20):
    A = _as_matrix(mat, 1)
    return CompressSparseMatrix(A, compressTol) if compress else A


def RestrictMatrixToBlocks(mat, rowBlocks, colBlocks=None, tol=1e-20):
    """keep the entries (i, j) for which some k has i in rowBlocks[k] and j in colBlocks[k] (default: same blocks)"""
    A = _as_matrix(mat, 1)
    colBlocks = rowBlocks if colBlocks is None else colBlocks
    rb = -np.ones(A.n_rows, dtype=np.int64)
    cb = -np.ones(A.n_cols, dtype=np.int64)
    for k, (r, c) in enumerate(zip(rowBlocks, colBlocks)):
        rb[np.asarray(list(r), dtype=np.int64)] = k
        cb[np.asarray(list(c), dtype=np.int64)] = k
    rows = np.repeat(np.arange(A.n_rows), np.diff(A.rowptr))
    keep = (rb[rows] >= 0) & (rb[rows] == cb[np.asarray(A.col)])
    bb = A.br * A.bc
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=A.n_rows))]).astype(np.int64)
    R = Matrix(A.n_rows, A.n_cols, A.br, A.bc, rp, np.asarray(A.col)[keep].astype(np.int32),
               np.asarray(A.val).reshape(-1, bb)[keep].ravel())
    return CompressSparseMatrix(R, tol)


def GetMemoryUse(mat):
    """bytes of the stored (block-)CSR arrays"""
    if mat is None:
        return 0
    A = _as_matrix(mat, 1)
    return int(np.asarray(A.rowptr).nbytes + np.asarray(A.col).nbytes + np.asarray(A.val).nbytes)


class ProxySmoother(BaseSmoother):
    """k-step / symmetric wrapper (reference base_smoother.hpp:169-229)"""

    def __init__(self, smoother, nsteps=1, symm=False):
        self._sm, self._n, self._symm = smoother, nsteps, symm

    def _symm_k(self, x, rhs, res, ru, ur, xz):
        res = self._sm._res(x, res)
        self._sm.Smooth(x, rhs, res, ru, ur, xz)
        self._sm.SmoothBack(x, rhs, res, ur, ur, False)
        for _ in range(self._n - 1):
            self._sm.Smooth(x, rhs, res, ur, ur, False)
            self._sm.SmoothBack(x, rhs, res, ur, ur, False)

    def Smooth(self, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        if self._symm:
            self._symm_k(x, rhs, res, res_updated, update_res, x_zero)
        else:
            self._sm.SmoothK(self._n, x, rhs, res, res_updated, update_res, x_zero)

    def SmoothBack(self, x, rhs, res=None, res_updated=False, update_res=False, x_zero=False):
        if self._symm:
            self._symm_k(x, rhs, res, res_updated, update_res, x_zero)
        else:
            self._sm.SmoothBackK(self._n, x, rhs, res, res_updated, update_res, x_zero)

    def GetMatrix(self):
        return self._sm.GetMatrix()
