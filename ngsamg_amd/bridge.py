"""Bridge from the reference's parallel layout to the owner-row layout of the MI355X path (host side, setup time).

NGSolve / NgsAMG distribute a finite element matrix by ELEMENTS: every rank assembles its own elements only, so a DOF
on a subdomain interface exists on every rank that touches it ("shared"), and its matrix row on each of these ranks holds
only that rank's partial sum -- the true operator is A = sum_r R_r^T A_r R_r (C2D ParallelMatrix; reference
src/base/precond/amg_pc.cpp:1054-1058, universal_dofs.cpp:194-218).  Every shared DOF has one MASTER, the lowest rank
that holds it (BasicDCCMap::CalcDOFMasters, src/base/linalg/dcc_map.cpp:497-543), and HybridMatrix re-sorts the entries
into M (master x master, with the other ranks' diagonal blocks added in) and G (src/base/linalg/hybrid_matrix.cpp:17-307).

The GPU path stores complete rows at the owner instead: owner = the master, columns [owned | ghost].
`from_shared_layout` converts: non-master ranks send their partial rows of a shared DOF to its master, which adds them up;
columns are re-indexed to (owner rank, index at the owner).  It also returns the vector map of the shared DOFs -- the
DCCMap tables m_ex_dofs / g_ex_dofs of the reference in the form amgx_halo_exchange takes -- so a DISTRIBUTED NGSolve
vector (partial sums on shared DOFs) becomes an owner vector by one ghost -> owner add, and an owner vector becomes a
CUMULATED NGSolve vector by one owner -> ghost copy.

Input per rank (what an NGSolve-side shim reads off ParallelDofs and the local SparseMatrix):
    A_loc       scipy CSR, n_loc x n_loc, the rank's partial sums
    dist_procs  list of n_loc arrays: the OTHER ranks sharing each local DOF (ParallelDofs::GetDistantProcs(dof))
    free, coords (optional) per local DOF
Convention used for the pairing of shared DOFs between two ranks p, q (NGSolve's GetExchangeDofs(q) tables): the k-th
shared DOF in ascending local order on p is the k-th on q.  Generators that cannot guarantee this pass `ex_key` (any
globally consistent sortable key per local DOF, e.g. a global vertex number); the lists are then ordered by it.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from ._lib import NgsAMGError


class SharedLocal:
    """one rank's data in the reference's layout"""

    def __init__(self, rank, A_loc, dist_procs, free=None, coords=None, ex_key=None):
        self.rank = int(rank)
        self.A = sp.csr_matrix(A_loc)
        self.n_loc = self.A.shape[0]
        self.dist_procs = [np.asarray(d, dtype=np.int64) for d in dist_procs]
        if len(self.dist_procs) != self.n_loc:
            raise NgsAMGError("SharedLocal: one dist_procs entry per local dof")
        self.free = np.ones(self.n_loc, dtype=np.uint8) if free is None else np.ascontiguousarray(free, dtype=np.uint8)
        self.coords = None if coords is None else np.ascontiguousarray(coords, dtype=np.float64)
        self.ex_key = None if ex_key is None else np.asarray(ex_key)


class VectorMap:
    """local (reference) numbering <-> owner layout of one rank.

    ext layout = [owned dofs | copies of shared dofs mastered elsewhere, grouped by master]; `perm[e]` = local dof at ext
    position e.  `halo` = (peers, send_ptr, send_idx, recv_ptr): the DCCMap tables of the shared dofs in the form of
    amgx_halo_desc -- send_idx lists, per peer, MY owned dofs that the peer holds a copy of."""

    def __init__(self, n_own, perm, peers, send_ptr, send_idx, recv_ptr):
        self.n_own, self.perm = int(n_own), np.asarray(perm, dtype=np.int64)
        self.peers, self.send_ptr, self.send_idx, self.recv_ptr = peers, send_ptr, send_idx, recv_ptr

    @property
    def n_ext(self):
        return self.perm.size

    def to_ext(self, v_loc):
        return np.ascontiguousarray(np.asarray(v_loc)[self.perm])

    def from_ext(self, v_ext, n_loc):
        out = np.zeros(n_loc, dtype=np.asarray(v_ext).dtype)
        out[self.perm] = v_ext
        return out


def _exchange_lists(L):
    """peer -> local dofs shared with that peer, ordered consistently on both sides"""
    ex = {}
    for d, procs in enumerate(L.dist_procs):
        for q in procs:
            ex.setdefault(int(q), []).append(d)
    out = {}
    for q, lst in ex.items():
        a = np.asarray(lst, dtype=np.int64)
        if L.ex_key is not None:
            a = a[np.argsort(L.ex_key[a], kind="stable")]
        out[q] = a
    return out


def from_shared_layout(comm, locs):
    """comm: LoopbackComm / TorchComm of ngsamg_amd.dist; locs: one SharedLocal per local rank.
    Returns (states, vmaps): RankState objects in owner-row form (input of DistributedAMG) and the VectorMap of each rank."""
    from .dist import RankState, _symmetrise
    ex = [_exchange_lists(L) for L in locs]
    # masters: lowest rank among the sharers (dcc_map.cpp:516-529)
    master, own_idx, n_own = [], [], []
    for L in locs:
        m = np.array([min([L.rank] + [int(q) for q in p]) for p in L.dist_procs], dtype=np.int64)
        master.append(m)
        mine = m == L.rank
        idx = np.full(L.n_loc, -1, dtype=np.int64)
        idx[mine] = np.arange(int(mine.sum()))
        own_idx.append(idx)
        n_own.append(int(mine.sum()))

    class _S:            # minimal duck type for _symmetrise
        def __init__(self, r):
            self.rank = r
    shims = [_S(L.rank) for L in locs]
    # 1. every master tells the sharers the index of the dof in its owned numbering
    sends = [{q: own_idx[i][lst] for q, lst in ex[i].items()} for i in range(len(locs))]
    recvs = comm.exchange(_symmetrise(comm, shims, sends, np.int64))
    for i, L in enumerate(locs):
        for q, lst in ex[i].items():
            got = recvs[i].get(q)
            if got is None or got.size != lst.size:
                raise NgsAMGError("shared-dof lists of two ranks do not pair up (pass ex_key)")
            sel = master[i][lst] == q
            own_idx[i][lst[sel]] = got[sel]
        if np.any(own_idx[i] < 0):
            raise NgsAMGError("a shared dof did not learn its index at the master")
    # 2. partial rows of non-master shared dofs travel to the master as (target index, owner of column, index of column, value)
    sends_i, sends_v = [], []
    for i, L in enumerate(locs):
        A = L.A
        rows = np.repeat(np.arange(L.n_loc), np.diff(A.indptr))
        di, dv = {}, {}
        away = master[i][rows] != L.rank
        for q in np.unique(master[i][rows[away]]):
            sel = away & (master[i][rows] == q)
            r, c = rows[sel], A.indices[sel]
            di[int(q)] = np.stack([own_idx[i][r], master[i][c], own_idx[i][c]], axis=1).astype(np.int64).ravel()
            dv[int(q)] = A.data[sel].astype(np.float64)
        sends_i.append(di)
        sends_v.append(dv)
    ri = comm.exchange(_symmetrise(comm, shims, sends_i, np.int64))
    rv = comm.exchange(_symmetrise(comm, shims, sends_v, np.float64))
    states, vmaps = [], []
    for i, L in enumerate(locs):
        A = L.A
        rows = np.repeat(np.arange(L.n_loc), np.diff(A.indptr))
        keep = master[i][rows] == L.rank
        tr = [own_idx[i][rows[keep]]]
        co = [master[i][A.indices[keep]]]
        ci = [own_idx[i][A.indices[keep]]]
        va = [A.data[keep]]
        for q, t in ri[i].items():
            if t.size:
                t = t.reshape(-1, 3)
                tr.append(t[:, 0]); co.append(t[:, 1]); ci.append(t[:, 2]); va.append(rv[i][q])
        tr, co, ci, va = (np.concatenate(x) for x in (tr, co, ci, va))
        # ghost columns: keys (owner, index) of columns owned elsewhere, sorted by (owner, index)
        gh = co != L.rank
        gkeys = np.unique(np.stack([co[gh], ci[gh]], axis=1), axis=0) if gh.any() else np.empty((0, 2), dtype=np.int64)
        col = np.where(gh, 0, ci)
        if gh.any():
            # position of every ghost key in the sorted unique list
            big = int(max(ci.max(), gkeys[:, 1].max())) + 1 if gkeys.size else 1
            pos = np.searchsorted(gkeys[:, 0] * big + gkeys[:, 1], co[gh] * big + ci[gh])
            col[gh] = n_own[i] + pos
        Aown = sp.coo_matrix((va, (tr, col)), shape=(n_own[i], n_own[i] + gkeys.shape[0])).tocsr()   # duplicates are summed
        Aown.sort_indices()
        st = RankState()
        st.rank, st.n = L.rank, n_own[i]
        st.A = Aown
        # the hybrid smoother's off-rank weight sums |partial g_kj| over the ranks (CalcHybridSmootherRDG iterates the local G
        # of every rank and all-reduces, hybrid_smoother_utils.hpp:74-103), which exceeds |assembled g_kj| where partial sums
        # of opposite sign meet: keep the sum of the absolute partial values of the ghost columns for _hybrid_gs_data
        if gkeys.shape[0]:
            Gabs = sp.coo_matrix((np.abs(va[gh]), (tr[gh], col[gh] - n_own[i])), shape=(n_own[i], gkeys.shape[0])).tocsr()
            Gabs.sort_indices()
            st.G_abs_partial = Gabs
        mine = np.nonzero(master[i] == L.rank)[0]
        st.free = np.ascontiguousarray(L.free[mine])
        st.coords = None if L.coords is None else np.ascontiguousarray(L.coords[mine])
        st.ghost_owner = gkeys[:, 0].astype(np.int64)
        st.ghost_rindex = gkeys[:, 1].astype(np.int64)
        states.append(st)
        # vector map of the shared dofs (DCCMap tables): ext = [owned | non-master shared dofs grouped by master]
        nm = np.nonzero(master[i] != L.rank)[0]
        order = np.lexsort((own_idx[i][nm], master[i][nm]))
        nm = nm[order]
        perm = np.concatenate([mine, nm])
        peers = sorted(ex[i].keys())
        send_ptr, recv_ptr, sidx = [0], [0], []
        for q in peers:
            lst = ex[i][q]
            mineq = lst[master[i][lst] == L.rank]
            # the peer orders its copies of my dofs by my index: send in that order
            mineq = mineq[np.argsort(own_idx[i][mineq], kind="stable")]
            sidx.append(own_idx[i][mineq])
            send_ptr.append(send_ptr[-1] + mineq.size)
            recv_ptr.append(recv_ptr[-1] + int((master[i][nm] == q).sum()))
        vmaps.append(VectorMap(n_own[i], perm, np.asarray(peers, dtype=np.int32), np.asarray(send_ptr, dtype=np.int64),
                               np.ascontiguousarray(np.concatenate(sidx) if sidx else np.empty(0), dtype=np.int32),
                               np.asarray(recv_ptr, dtype=np.int64)))
    return states, vmaps


def accumulate_host(comm, vmaps, v_exts):
    """host reference of the ghost -> owner add on the shared-dof map (DCCMap DIS2CO): returns the owned parts"""
    from .dist import _symmetrise

    class _S:
        def __init__(self, r):
            self.rank = r
    # ranks are identified by position in comm.local_ranks for loopback, by comm.rank otherwise
    ranks = list(getattr(comm, "local_ranks", [getattr(comm, "rank", 0)]))
    shims = [_S(r) for r in ranks]
    sends = []
    for vm, v in zip(vmaps, v_exts):
        sends.append({int(q): np.asarray(v[vm.n_own + vm.recv_ptr[k]:vm.n_own + vm.recv_ptr[k + 1]], dtype=np.float64) for k, q in enumerate(vm.peers)})
    recvs = comm.exchange(_symmetrise(comm, shims, sends, np.float64))
    outs = []
    for vm, v, r in zip(vmaps, v_exts, recvs):
        o = np.array(v[:vm.n_own], dtype=np.float64)
        for k, q in enumerate(vm.peers):
            idx = vm.send_idx[vm.send_ptr[k]:vm.send_ptr[k + 1]]
            got = r.get(int(q), np.empty(0))
            if got.size != idx.size:
                raise NgsAMGError("vector map: size mismatch")
            np.add.at(o, idx, got)
        outs.append(o)
    return outs


# ------------------------------------------------------------------------------------------------------------
# synthetic generator of the reference's layout (tests, examples): structured Kuhn grid, cells split into boxes
# ------------------------------------------------------------------------------------------------------------

def shared_poisson_partition(rank, pgrid, gshape, dirichlet="right|top", jitter=0.2, seed=1):
    """The rank's data as an NGSolve rank would hold it for the P1 Poisson problem on a structured grid of `gshape`
    vertices whose CELLS are split into a pgrid box of ranks: local dofs = the vertices of the rank's cells (interface
    planes duplicated), local matrix = the rank's own cells only (partial sums on the interface)."""
    import ctypes as C
    from . import _lib
    from .dist import hashed_coords
    dim = len(pgrid)
    pc = np.unravel_index(rank, pgrid)
    lo, hi = [], []
    for d in range(dim):
        cells = gshape[d] - 1
        cuts = [(cells * k) // pgrid[d] for k in range(pgrid[d] + 1)]
        lo.append(cuts[pc[d]])
        hi.append(cuts[pc[d] + 1] + 1)           # vertices [lo, hi)
    shape = tuple(hi[d] - lo[d] for d in range(dim))
    X = hashed_coords(lo, hi, gshape, jitter, seed)
    n = int(np.prod(shape))
    coords = np.ascontiguousarray(X.reshape(n, dim))
    lib = _lib.host()
    shp = np.asarray(shape, dtype=np.int64)
    rowptr = np.empty(n + 1, dtype=np.int64)
    _lib.hcheck(lib.amgh_kuhn_pattern(dim, _lib.ptr(shp, C.c_int64), _lib.ptr(rowptr, C.c_int64)))
    col = np.empty(int(rowptr[-1]), dtype=np.int32)
    val = np.empty(int(rowptr[-1]))
    _lib.hcheck(lib.amgh_kuhn_assemble(dim, _lib.ptr(shp, C.c_int64), _lib.ptr(coords, C.c_double), 0, 1, 1.0, 0.0, None,
                                       _lib.ptr(rowptr, C.c_int64), _lib.ptr(col, C.c_int32), _lib.ptr(val, C.c_double), None))
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))
    idx = np.stack(np.unravel_index(np.arange(n), shape), axis=1) + np.asarray(lo)
    gid = np.ravel_multi_index(tuple(idx.T), gshape)
    # sharing sets from geometry: a vertex belongs to every rank whose vertex box contains it
    boxes = []
    for r in range(int(np.prod(pgrid))):
        p = np.unravel_index(r, pgrid)
        b = []
        for d in range(dim):
            cells = gshape[d] - 1
            b.append(((cells * p[d]) // pgrid[d], (cells * (p[d] + 1)) // pgrid[d]))       # inclusive vertex range
        boxes.append(b)
    inside = np.ones((len(boxes), n), dtype=bool)
    for r, b in enumerate(boxes):
        for d in range(dim):
            inside[r] &= (idx[:, d] >= b[d][0]) & (idx[:, d] <= b[d][1])
    dist_procs = [np.nonzero(inside[:, k])[0] for k in range(n)]
    dist_procs = [p[p != rank] for p in dist_procs]
    free = np.ones(n, dtype=np.uint8)
    names = {"left": (0, 0), "right": (0, gshape[0] - 1), "bottom": (dim - 1, 0), "top": (dim - 1, gshape[dim - 1] - 1)}
    if dim == 3:
        names.update({"front": (1, 0), "back": (1, gshape[1] - 1)})
    for nm in (dirichlet.split("|") if dirichlet else []):
        ax, v = names[nm]
        free[idx[:, ax] == v] = 0
    return SharedLocal(rank, A, dist_procs, free=free, coords=coords, ex_key=gid), gid
