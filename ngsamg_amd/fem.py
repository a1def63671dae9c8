"""Synthetic P1 finite-element matrices on structured simplicial meshes.

This is the stand-in for the *caller* of the reference (NGSolve + netgen, which
are not available): it plays the role of ``setup_poisson`` / ``setup_elast`` of
the reference's test drivers (reference tests/h1/amg_utils.py:122-131,
tests/elasticity/amg_utils.py:333-435) and produces what NGSolve would hand to
the preconditioner: a (block-)CSR matrix with the element-connectivity sparsity
pattern, sorted int32 columns, row-major fp64 blocks, and a free-dof mask
(Dirichlet rows are *kept* and masked, reference gssmoother.cpp:151-167).

Meshes: unit square split into right triangles, unit cube split into 6 Kuhn
tetrahedra per cell.  Interior vertices are jittered (seeded) so that every
stored entry is non-zero, as on the netgen meshes the reference is tested on
(SURVEY.md section 8d).

Everything is vectorised over stencil offsets: for a simplex type with local
vertices at cell offsets o_a, the contribution K_ab of all cells is a plain
slice-add into the array that stores A[v, v + (o_b - o_a)].
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field

import numpy as np


@dataclass
class FEMProblem:
    """A (block-)CSR FEM operator plus what the preconditioner needs to know."""
    n: int                     # number of block rows (vertices)
    bs: int                    # block size (dofs per vertex)
    rowptr: np.ndarray         # int64 [n+1]
    col: np.ndarray            # int32 [nnz]  ascending per row
    val: np.ndarray            # float64 [nnz, bs, bs] (row-major blocks) or [nnz] for bs == 1
    free: np.ndarray           # uint8 [n]  1 = free vertex (all dofs of the vertex share it)
    coords: np.ndarray         # float64 [n, dim]
    dim: int
    shape: tuple               # vertices per direction
    load: np.ndarray = field(default=None)   # float64 [n*bs] consistent load vector for f = 1 (masked)

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    @property
    def ndof(self):
        return self.n * self.bs

    def to_scipy(self):
        import scipy.sparse as sp
        if self.bs == 1:
            return sp.csr_matrix((self.val, self.col, self.rowptr), shape=(self.n, self.n))
        return sp.bsr_matrix((self.val, self.col, self.rowptr),
                             shape=(self.n * self.bs, self.n * self.bs)).tocsr()


# ----------------------------------------------------------------------------------------------
# mesh topology helpers
# ----------------------------------------------------------------------------------------------

def _kuhn_simplices(dim):
    """Local vertex offsets of the dim! Kuhn simplices of the unit cell."""
    simplices = []
    for perm in itertools.permutations(range(dim)):
        v = np.zeros(dim, dtype=np.int64)
        verts = [v.copy()]
        for ax in perm:
            v[ax] += 1
            verts.append(v.copy())
        simplices.append(np.array(verts))
    return simplices


def _stencil_offsets(dim):
    """All offsets o_b - o_a occurring in Kuhn simplices, sorted by linear index delta (z fastest)."""
    offs = set()
    for s in _kuhn_simplices(dim):
        for a in range(dim + 1):
            for b in range(dim + 1):
                offs.add(tuple(int(t) for t in (s[b] - s[a])))
    return offs


def _jittered_coords(shape, jitter, seed):
    dim = len(shape)
    axes = [np.linspace(0.0, 1.0, s) for s in shape]
    grid = np.stack(np.meshgrid(*axes, indexing="ij"), axis=-1)  # shape + (dim,)
    if jitter > 0.0:
        rng = np.random.default_rng(seed)
        pert = rng.uniform(-jitter, jitter, size=grid.shape)
        for d in range(dim):
            h = 1.0 / (shape[d] - 1)
            pert[..., d] *= h
        # only interior vertices move: the boundary stays the unit square / cube
        interior = np.ones(shape, dtype=bool)
        for d in range(dim):
            sl = [slice(None)] * dim
            sl[d] = 0
            interior[tuple(sl)] = False
            sl[d] = shape[d] - 1
            interior[tuple(sl)] = False
        grid = grid + pert * interior[..., None]
    return grid


def _simplex_gradients(X):
    """X: [..., dim+1, dim] vertex coordinates.  Returns (vol [...], grads [..., dim+1, dim])."""
    dim = X.shape[-1]
    E = X[..., 1:, :] - X[..., :1, :]          # rows = edge vectors
    if dim == 2:
        a, b = E[..., 0, 0], E[..., 0, 1]
        c, d = E[..., 1, 0], E[..., 1, 1]
        det = a * d - b * c
        inv = np.empty_like(E)
        inv[..., 0, 0] = d / det
        inv[..., 0, 1] = -b / det
        inv[..., 1, 0] = -c / det
        inv[..., 1, 1] = a / det
        vol = np.abs(det) / 2.0
    else:
        a = E
        c00 = a[..., 1, 1] * a[..., 2, 2] - a[..., 1, 2] * a[..., 2, 1]
        c01 = a[..., 1, 2] * a[..., 2, 0] - a[..., 1, 0] * a[..., 2, 2]
        c02 = a[..., 1, 0] * a[..., 2, 1] - a[..., 1, 1] * a[..., 2, 0]
        det = a[..., 0, 0] * c00 + a[..., 0, 1] * c01 + a[..., 0, 2] * c02
        inv = np.empty_like(E)
        inv[..., 0, 0] = c00 / det
        inv[..., 1, 0] = c01 / det
        inv[..., 2, 0] = c02 / det
        inv[..., 0, 1] = (a[..., 0, 2] * a[..., 2, 1] - a[..., 0, 1] * a[..., 2, 2]) / det
        inv[..., 1, 1] = (a[..., 0, 0] * a[..., 2, 2] - a[..., 0, 2] * a[..., 2, 0]) / det
        inv[..., 2, 1] = (a[..., 0, 1] * a[..., 2, 0] - a[..., 0, 0] * a[..., 2, 1]) / det
        inv[..., 0, 2] = (a[..., 0, 1] * a[..., 1, 2] - a[..., 0, 2] * a[..., 1, 1]) / det
        inv[..., 1, 2] = (a[..., 0, 2] * a[..., 1, 0] - a[..., 0, 0] * a[..., 1, 2]) / det
        inv[..., 2, 2] = (a[..., 0, 0] * a[..., 1, 1] - a[..., 0, 1] * a[..., 1, 0]) / det
        vol = np.abs(det) / 6.0
    # lambda_k(x) = (E^-1 column k) . (x - x0)  for k = 1..dim ;  grad lambda_k = inv[:, k-1]
    g = np.empty(X.shape, dtype=X.dtype)
    g[..., 1:, :] = np.swapaxes(inv, -1, -2)
    g[..., 0, :] = -g[..., 1:, :].sum(axis=-2)
    return vol, g


# ----------------------------------------------------------------------------------------------
# stencil accumulation -> CSR
# ----------------------------------------------------------------------------------------------

class _StencilAccumulator:
    """Holds A[v, v+off] for all vertices v as one array per stencil offset."""

    def __init__(self, shape, bs):
        self.shape = tuple(shape)
        self.dim = len(shape)
        self.bs = bs
        strides = [1] * self.dim
        for d in range(self.dim - 2, -1, -1):
            strides[d] = strides[d + 1] * shape[d + 1]
        self.strides = strides
        offs = sorted(_stencil_offsets(self.dim), key=lambda o: sum(o[d] * strides[d] for d in range(self.dim)))
        self.offsets = offs
        blk = () if bs == 1 else (bs, bs)
        self.data = {o: np.zeros(self.shape + blk) for o in offs}

    def add(self, o_a, o_b, cell_slices, K):
        """K: [cells..., (bs, bs)] contribution to entry (v+o_a, v+o_b) for the cells in cell_slices."""
        off = tuple(int(o_b[d] - o_a[d]) for d in range(self.dim))
        sl = tuple(slice(cs.start + int(o_a[d]), cs.stop + int(o_a[d])) for d, cs in enumerate(cell_slices))
        self.data[off][sl] += K

    def to_csr(self):
        shape, dim, bs = self.shape, self.dim, self.bs
        n = int(np.prod(shape))
        idx = np.arange(n, dtype=np.int64).reshape(shape)
        valid = []
        for o in self.offsets:
            m = np.ones(shape, dtype=bool)
            for d in range(dim):
                sl = [slice(None)] * dim
                if o[d] > 0:
                    sl[d] = slice(shape[d] - o[d], None)
                    m[tuple(sl)] = False
                elif o[d] < 0:
                    sl[d] = slice(0, -o[d])
                    m[tuple(sl)] = False
            valid.append(m.reshape(-1))
        counts = np.zeros(n, dtype=np.int64)
        for m in valid:
            counts += m
        rowptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(counts, out=rowptr[1:])
        nnz = int(rowptr[-1])
        col = np.empty(nnz, dtype=np.int32)
        val = np.empty((nnz,) if bs == 1 else (nnz, bs, bs))
        pos = rowptr[:-1].copy()
        flat_idx = idx.reshape(-1)
        for o, m in zip(self.offsets, valid):
            delta = sum(o[d] * self.strides[d] for d in range(dim))
            p = pos[m]
            col[p] = (flat_idx[m] + delta).astype(np.int32)
            arr = self.data[o].reshape((n,) if bs == 1 else (n, bs, bs))
            val[p] = arr[m]
            pos[m] += 1
        return rowptr, col, val


def _boundary_mask(shape, dirichlet):
    """free mask from face names: left/right = x min/max, bottom/top = last axis min/max,
    front/back = y min/max (3D only)."""
    dim = len(shape)
    free = np.ones(shape, dtype=np.uint8)
    names = {"left": (0, 0), "right": (0, -1), "bottom": (dim - 1, 0), "top": (dim - 1, -1)}
    if dim == 3:
        names.update({"front": (1, 0), "back": (1, -1)})
    if dirichlet in (None, ""):
        return free.reshape(-1)
    for name in dirichlet.split("|"):
        if name == ".*":
            for d in range(dim):
                sl = [slice(None)] * dim
                sl[d] = 0
                free[tuple(sl)] = 0
                sl[d] = -1
                free[tuple(sl)] = 0
            continue
        ax, side = names[name]
        sl = [slice(None)] * dim
        sl[ax] = side
        free[tuple(sl)] = 0
    return free.reshape(-1)


def _slabs(ncell0, chunk):
    s = 0
    while s < ncell0:
        e = min(ncell0, s + chunk)
        yield slice(s, e)
        s = e


def _cell_vertex_coords(grid, simplex, cell_slices):
    dim = grid.ndim - 1
    X = []
    for a in range(dim + 1):
        sl = tuple(slice(cs.start + int(simplex[a][d]), cs.stop + int(simplex[a][d])) for d, cs in enumerate(cell_slices))
        X.append(grid[sl])
    return np.stack(X, axis=-2)     # [cells..., dim+1, dim]


# ----------------------------------------------------------------------------------------------
# public generators
# ----------------------------------------------------------------------------------------------

def poisson(shape, dirichlet="right|top", jitter=0.2, seed=1, coef=None, chunk=16):
    """P1 stiffness matrix of -div(coef grad u) on the unit square / cube.

    shape: vertices per direction, e.g. (224, 224) or (215, 215, 215).
    coef:  optional callable(centroids [..., dim]) -> coefficient per cell (coefficient jumps,
           reference tests/h1/jump/test_2d_jump_lo.py).
    """
    shape = tuple(int(s) for s in shape)
    dim = len(shape)
    grid = _jittered_coords(shape, jitter, seed)
    acc = _StencilAccumulator(shape, 1)
    load = np.zeros(shape)
    simplices = _kuhn_simplices(dim)
    for s0 in _slabs(shape[0] - 1, chunk):
        cell_slices = (s0,) + tuple(slice(0, shape[d] - 1) for d in range(1, dim))
        for simplex in simplices:
            X = _cell_vertex_coords(grid, simplex, cell_slices)
            vol, g = _simplex_gradients(X)
            w = vol
            if coef is not None:
                w = vol * coef(X.mean(axis=-2))
            Kall = np.matmul(g, np.swapaxes(g, -1, -2)) * w[..., None, None]   # [cells..., dim+1, dim+1]
            for a in range(dim + 1):
                sl = tuple(slice(cs.start + int(simplex[a][d]), cs.stop + int(simplex[a][d])) for d, cs in enumerate(cell_slices))
                load[sl] += vol / (dim + 1)
                for b in range(dim + 1):
                    acc.add(simplex[a], simplex[b], cell_slices, Kall[..., a, b])
    rowptr, col, val = acc.to_csr()
    free = _boundary_mask(shape, dirichlet)
    n = int(np.prod(shape))
    return FEMProblem(n=n, bs=1, rowptr=rowptr, col=col, val=val, free=free,
                      coords=grid.reshape(n, dim), dim=dim, shape=shape,
                      load=load.reshape(-1) * free)


def elasticity(shape, dirichlet="left", mu=1.0, lam=0.0, jitter=0.2, seed=1, rotations=False, chunk=8,
               extent=None):
    """P1 linear elasticity on the unit square / cube (or a box of the given extent).

    rotations=False: displacement formulation  mu*(eps(u),eps(v)) + lam*(div u, div v)   (bs = dim),
        as reference tests/elasticity/amg_utils.py:333-361 (setup_norot_elast).
    rotations=True:  displacement + rotation dofs  mu*(grad u - skew(w), grad v - skew(z)) + lam*div*div
        (bs = 3 in 2D, 6 in 3D), as setup_rot_elast (:364-421).
    """
    shape = tuple(int(s) for s in shape)
    dim = len(shape)
    nrot = (dim * (dim - 1)) // 2
    bs = dim + nrot if rotations else dim
    grid = _jittered_coords(shape, jitter, seed)
    if extent is not None:
        grid = grid * np.asarray(extent, dtype=float)
    acc = _StencilAccumulator(shape, bs)
    load = np.zeros(shape + (bs,))
    simplices = _kuhn_simplices(dim)
    eye = np.eye(dim)
    # skew basis: w -> skew(w);  S[r] = d skew / d w_r  (dim x dim)
    if dim == 2:
        S = np.array([[[0.0, -1.0], [1.0, 0.0]]])
    else:
        S = np.zeros((3, 3, 3))
        S[0] = [[0, 0, 0], [0, 0, -1], [0, 1, 0]]
        S[1] = [[0, 0, 1], [0, 0, 0], [-1, 0, 0]]
        S[2] = [[0, -1, 0], [1, 0, 0], [0, 0, 0]]
    SS = np.einsum("rij,sij->rs", S, S)   # <S_r, S_s>_F
    for s0 in _slabs(shape[0] - 1, chunk):
        cell_slices = (s0,) + tuple(slice(0, shape[d] - 1) for d in range(1, dim))
        for simplex in simplices:
            X = _cell_vertex_coords(grid, simplex, cell_slices)
            vol, g = _simplex_gradients(X)
            for a in range(dim + 1):
                sl = tuple(slice(cs.start + int(simplex[a][d]), cs.stop + int(simplex[a][d])) for d, cs in enumerate(cell_slices))
                load[sl + (dim - 1,)] += -vol / (dim + 1)     # unit body force in the last displacement direction
                ga = g[..., a, :]
                for b in range(dim + 1):
                    gb = g[..., b, :]
                    gagb = np.einsum("...i,...i->...", ga, gb)
                    K = np.zeros(vol.shape + (bs, bs))
                    if rotations:
                        # u-u: mu * (grad u_b , grad v_a) = mu * (ga.gb) I
                        K[..., :dim, :dim] = mu * gagb[..., None, None] * eye
                        # u-w / w-u:  -mu * <phi_b skew(e_r), e_i (x) ga>  with P1 mass-type integrals
                        # int phi_b dx over simplex = vol/(dim+1)
                        m = 1.0 / (dim + 1)
                        # test v = e_i phi_a (grad v = e_i ga^T), trial w_r phi_b:
                        #   -mu * int <phi_b S_r, e_i ga^T> = -mu * m * (S_r ga)_i
                        Sg_a = np.einsum("rij,...j->...ir", S, ga)        # [..., i, r]
                        Sg_b = np.einsum("rij,...j->...ir", S, gb)
                        K[..., :dim, dim:] = -mu * m * Sg_a
                        # test z_r phi_a, trial u = e_j phi_b:  -mu * m * (S_r gb)_j
                        K[..., dim:, :dim] = -mu * m * np.swapaxes(Sg_b, -1, -2)
                        # w-w: mu * int phi_a phi_b <S_r,S_s> ;  int phi_a phi_b = vol (1+delta_ab)/((dim+1)(dim+2))
                        mab = (2.0 if a == b else 1.0) / ((dim + 1) * (dim + 2))
                        K[..., dim:, dim:] = mu * mab * SS
                    else:
                        # mu*(eps(u),eps(v)) = mu/2 * [ (ga.gb) I + gb ga^T ]
                        K[..., :dim, :dim] = 0.5 * mu * (gagb[..., None, None] * eye
                                                         + gb[..., :, None] * ga[..., None, :])
                    if lam != 0.0:
                        K[..., :dim, :dim] += lam * ga[..., :, None] * gb[..., None, :]
                    K *= vol[..., None, None]
                    acc.add(simplex[a], simplex[b], cell_slices, K)
    rowptr, col, val = acc.to_csr()
    free = _boundary_mask(shape, dirichlet)
    n = int(np.prod(shape))
    load = load.reshape(n, bs) * free[:, None]
    return FEMProblem(n=n, bs=bs, rowptr=rowptr, col=col, val=val, free=free,
                      coords=grid.reshape(n, dim), dim=dim, shape=shape,
                      load=load.reshape(-1))


# ----------------------------------------------------------------------------------------------
# fast path: the same matrices from the C++ row-gather assembler (libngsamg_host, amgh_kuhn_*)
# ----------------------------------------------------------------------------------------------

def _fast(shape, kind, bs, dirichlet, jitter, seed, mu, lam, coef, extent):
    import ctypes as C
    from . import _lib
    lib = _lib.host()
    shape = tuple(int(s) for s in shape)
    dim = len(shape)
    grid = _jittered_coords(shape, jitter, seed)
    if extent is not None:
        grid = grid * np.asarray(extent, dtype=float)
    n = int(np.prod(shape))
    coords = np.ascontiguousarray(grid.reshape(n, dim))
    shp = np.asarray(shape, dtype=np.int64)
    rowptr = np.empty(n + 1, dtype=np.int64)
    _lib.hcheck(lib.amgh_kuhn_pattern(dim, _lib.ptr(shp, C.c_int64), _lib.ptr(rowptr, C.c_int64)))
    nnz = int(rowptr[-1])
    col = np.empty(nnz, dtype=np.int32)
    val = np.empty(nnz * bs * bs, dtype=np.float64)
    load = np.empty(n * bs, dtype=np.float64)
    cc = None
    if coef is not None:
        cshape = tuple(s - 1 for s in shape)
        cen = 0.0
        for corner in itertools.product((0, 1), repeat=dim):
            sl = tuple(slice(c, c + cs) for c, cs in zip(corner, cshape))
            cen = cen + grid[sl]
        cc = np.ascontiguousarray(coef(cen / (1 << dim)), dtype=np.float64).reshape(-1)
    _lib.hcheck(lib.amgh_kuhn_assemble(dim, _lib.ptr(shp, C.c_int64), _lib.ptr(coords, C.c_double), kind, bs,
                                       float(mu), float(lam), _lib.ptr(cc, C.c_double), _lib.ptr(rowptr, C.c_int64),
                                       _lib.ptr(col, C.c_int32), _lib.ptr(val, C.c_double), _lib.ptr(load, C.c_double)))
    free = _boundary_mask(shape, dirichlet)
    load = (load.reshape(n, bs) * free[:, None]).reshape(-1)
    return FEMProblem(n=n, bs=bs, rowptr=rowptr, col=col, val=val if bs == 1 else val.reshape(nnz, bs, bs), free=free,
                      coords=coords, dim=dim, shape=shape, load=load)


def poisson_fast(shape, dirichlet="right|top", jitter=0.2, seed=1, coef=None):
    """Same operator as poisson() (coefficients per *cell* rather than per simplex), assembled in C++."""
    return _fast(shape, 0, 1, dirichlet, jitter, seed, 1.0, 0.0, coef, None)


def elasticity_fast(shape, dirichlet="left", mu=1.0, lam=0.0, jitter=0.2, seed=1, rotations=False, extent=None, coef=None):
    """coef (optional): callable of the cell centres -> factor on the cell's stiffness (material jumps, as for poisson_fast)"""
    dim = len(shape)
    bs = dim + (dim * (dim - 1)) // 2 if rotations else dim
    return _fast(shape, 2 if rotations else 1, bs, dirichlet, jitter, seed, mu, lam, coef, extent)
