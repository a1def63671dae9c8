#!/usr/bin/env python3
"""bench.py -- V-cycle applications per second of the MI355X-native NgsAMG apply path.

Workload (BASELINE.json configs[1], SURVEY.md 8d "cfg 2"): 3D P1 Poisson on the unit cube, 215^3 vertices
(9 938 375 DOF, ~15 nnz/row), jittered Kuhn tetrahedra (seed 1), Dirichlet on right|top, Jacobi smoother
(omega 0.9), V(1,1), max_coarse_size 50.  One "step" = one preconditioner application x = C b
(amgx_apply, i.e. BaseAMGPC::Mult of the reference) with b and x resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 (one rank per GPU; started by torch.distributed.run or, without a launcher, by this script itself): STRONG scaling
of the same problem -- BASELINE.json's metric is "3D H1 ~10M DOF at 1/2/4/8 GPU": the 215^3 grid (the very matrix of the
N = 1 run) is cut into N slabs of owned rows (ngsamg_amd/dist.py, balanced cuts), value = applications of the GLOBAL
preconditioner per second = steps / time.  The data path is native: rank-partitioned fine levels [interior | boundary],
halo pack kernels + ncclSend / ncclRecv on a communication stream behind the C ABI (amgx_dist_apply), one ncclAllGather
of the first replicated level, the coarse hierarchy replicated; torch.distributed (gloo) only carries the rendezvous, the
128-byte RCCL id, the host-side setup messages and the barriers of the timed region.  --scaling weak: one 215^3 box per
rank instead (value = ranks x steps / time); --config cfg4: the 342^3 = 40M-DOF grid over a box of ranks.

Prints ONE JSON line on rank 0 (contract of the driver) with the extra objects "roofline" (dominant kernel: the fused
level-0 pre-smoothing + residual + restriction pass, HIP-event timed inside the running cycle) and "cpu_baseline" (the
CPU oracle = restatement of the reference's cycle, timed on this box's host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


_RESULT_FD = None


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def emit(line):
    """the one result line, on the process's original stdout"""
    data = (line.rstrip("\n") + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def sp_nnz(A):
    """stored (block) entries of a level matrix held as scipy matrix (scalar-expanded blocks count once per scalar entry)"""
    return A.nnz if hasattr(A, "nnz") else int(A.rowptr[-1])


def run_distributed(args, torch, dist, world, rank, device, nv):
    """N > 1: rank-partitioned V-cycle, RCCL through the C ABI.  strong: the nv^3 grid of the N = 1 run cut into N pieces;
    weak: one nv^3 box per rank."""
    import ctypes as C
    from ngsamg_amd import _lib, dist as D
    from ngsamg_amd.device import matrix_bytes, vcycle_bytes
    t0 = time.time()
    comm = D.TorchComm()
    # slabs along the slowest axis: every rank keeps a full nv^3 box (weak scaling) and has at most 2 neighbours with one
    # nv^2 face each, instead of up to 7 neighbours (3 faces + 3 edges + 1 corner) in a 2 x 2 x 2 arrangement: fewer and
    # smaller point-to-point messages per halo exchange, fewer ghost columns.  NGSAMG_PGRID=box: cfg 4's arrangement
    pg = D.proc_grid(world, 3) if os.environ.get("NGSAMG_PGRID") == "box" else (world, 1, 1)
    elast = args.config in ("cfg3", "cfg5")
    strong = args.scaling == "strong"
    hier_kw = {"spw": 1} if args.hierarchy == "spw" else {"spw": 0}       # same agglomeration rule as the single-GPU line (see main)
    # strong: ONE global grid of nv^3 vertices (the matrix of the single-GPU run: same positions, same assembly), balanced cuts
    gsh, cmode = ((nv, nv, nv), "rng") if strong else (None, "hash")
    # levels stay rank-partitioned while every rank still has this many rows; strong scaling keeps level 1 (313 k rows in
    # total at cfg 2) partitioned, the first replicated level is then the 20 k-row level 2 instead of level 1
    dmin = args.dist_min_rows if args.dist_min_rows else ((5000 if strong else 20000) if elast else (10000 if strong else 50000))
    if elast:
        rot = args.config == "cfg5"
        st = D.assemble_elasticity_owned(rank, pg, (nv, nv, nv), rotations=rot, mu=1.0, lam=0.5, dirichlet="left", jitter=0.2, seed=1,
                                         gshape=gsh, coords=cmode)
        t1 = time.time()
        torch.cuda.synchronize()
        mem0 = torch.cuda.mem_get_info(device)[0]
        amg = D.DistributedAMG(comm, [st], dim=3, omega=0.9, dist_min_rows=dmin, device=device, max_coarse_size=50, max_levels=10,
                               energy=1, regularize_cmats=0 if rot else 1, sm_type={"jacobi": "jacobi", "gs": "hgs", "gs_mc": "gs"}[args.smoother], **hier_kw)
    else:
        st = D.assemble_poisson_owned(rank, pg, (nv, nv, nv), dirichlet="right|top", jitter=0.2, seed=1, gshape=gsh, coords=cmode)
        t1 = time.time()
        torch.cuda.synchronize()
        mem0 = torch.cuda.mem_get_info(device)[0]
        amg = D.DistributedAMG(comm, [st], dim=3, omega=0.9, dist_min_rows=dmin, device=device, max_coarse_size=50, max_levels=10,
                               sm_type={"jacobi": "jacobi", "gs": "hgs", "gs_mc": "gs"}[args.smoother], **hier_kw)
    bs0 = int(getattr(st, "bs", 1))
    torch.cuda.synchronize()
    hier_bytes = int(mem0 - torch.cuda.mem_get_info(device)[0])       # per rank: hierarchy copies, halo buffers, RCCL workspace
    free_s = np.repeat(st.free, bs0).astype(np.float64)
    t2 = time.time()
    lib = _lib.hip()
    kind, nr, rk = C.c_int32(), C.c_int32(), C.c_int32()
    lib.amgx_comm_info(amg._dev._comm, C.byref(kind), C.byref(nr), C.byref(rk), None)
    if (kind.value, nr.value) != (_lib.AMGX_COMM_RCCL, world):
        raise SystemExit(f"rank {rank}: the RCCL communicator has {nr.value} ranks, expected {world}")
    if rank == 0:
        log(f"[rank 0] owned-row assembly {t1 - t0:.1f}s, distributed setup + upload {t2 - t1:.1f}s; "
            f"distributed levels {amg.k}, sizes {[lv[0].n for lv in amg.dist_levels]}, interior {[lv[0].n_interior for lv in amg.dist_levels]}, "
            f"ghosts {[lv[0].ghost_owner.size for lv in amg.dist_levels]}, "
            f"replicated tail: {amg.tail_hier.n_levels} levels from n = {amg.tail_hier.levels[0].n}")
    # algorithmic bytes per rank and cycle: distributed levels (same per-level formula as the serial model) + replicated tail
    per_rank = 0
    for l in range(amg.k):
        L = amg.tops[0].levels[l]
        n = L.A.n_rows
        nc = amg.tops[0].levels[l + 1].A.n_rows
        bl, bcl = L.A.br, amg.tops[0].levels[l + 1].A.br
        per_rank += 2 * matrix_bytes(L.A) + matrix_bytes(L.P) + matrix_bytes(L.PT) + 16 * bl * bl * n + 15 * 8 * bl * n + 2 * 8 * bcl * nc
    per_rank += vcycle_bytes(amg.tail_hier)[0]
    rng = np.random.default_rng(rank)
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        try:
            b = amg.rhs_buffer(0)         # resident in the [owned | ghost] layout: no per-apply copy of b
        except Exception as e:            # (torch without __cuda_array_interface__ support: one device copy per apply)
            log(f"rhs_buffer unavailable ({e!r}); b is copied into the halo layout every apply")
            b = torch.empty(st.n * bs0, dtype=torch.float64, device=f"cuda:{device}")
        b.copy_(torch.from_numpy(rng.standard_normal(st.n * bs0) * free_s))
        x = torch.empty(st.n * bs0, dtype=torch.float64, device=f"cuda:{device}")
        for _ in range(args.warmup):
            amg.Mult([b], [x])
        stream.synchronize()
        ex0 = amg._dev.n_exchanges()
        dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            amg.Mult([b], [x])
        torch.cuda.synchronize()
        dist.barrier()
        te = time.perf_counter()
        ex_per_cycle = (amg._dev.n_exchanges() - ex0) / max(1, args.steps)
        xn_loc = float(torch.dot(x, x).item())
    tt = torch.tensor([te - ts], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())
    xn = torch.tensor([xn_loc], dtype=torch.float64)
    dist.all_reduce(xn)
    ms_per_step = 1e3 * elapsed / args.steps
    # the dominant kernel timed INSIDE the collective cycle (amgx_dist_time_kernel: every rank runs whole cycles with direct
    # launches, HIP events around the level-0 launch over the interior rows, the halo exchange in flight beside it)
    k_probe, k_in_cycle, k_op = None, False, (8 if args.smoother == "jacobi" and not elast else 9 if args.smoother == "gs" else 0)
    if k_op:
        ok = 1
        try:
            with torch.cuda.stream(stream):
                k_probe = amg._dev.time_kernel(0, k_op, reps=20)
                k_in_cycle = True
        except Exception as e:       # (a rank without such a kernel raises before the first collective: all ranks agree below)
            ok, k_probe = 0, None
            if rank == 0:
                log(f"in-cycle kernel timing unavailable: {e!r}")
        okt = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if not int(okt.item()):
            k_probe, k_in_cycle = None, False
    # PCG to 1e-8 through the rank-partitioned preconditioner (amgx_dist_pcg): iteration count of THIS hierarchy
    dpcg = None
    try:
        with torch.cuda.stream(stream):
            bb = b[:st.n * bs0].clone()
            xx = torch.zeros_like(bb)
            its, errs = amg.pcg([bb], [xx], tol=1e-8, maxsteps=200)
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            xx.zero_()
            its, errs = amg.pcg([bb], [xx], tol=1e-8, maxsteps=200)
            torch.cuda.synchronize()
            dpcg = {"tol": 1e-8, "iterations": int(its), "solve_ms": round(1e3 * (time.perf_counter() - t_s), 3),
                    "rel_err_estimate": float(errs[-1] / errs[0]) if len(errs) and errs[0] else None}
    except Exception as e:
        if rank == 0:
            log(f"distributed PCG failed: {e!r}")
    dist.barrier()
    # the hierarchy the ranks built together: global level sizes / entries, operator complexity
    lvl_n = torch.tensor([lv[0].n for lv in amg.dist_levels], dtype=torch.int64)
    lvl_nnz = torch.tensor([int(sp_nnz(lv[0].A)) // max(1, int(getattr(lv[0], "bs", 1))) ** 2 for lv in amg.dist_levels], dtype=torch.int64)
    dist.all_reduce(lvl_n)
    dist.all_reduce(lvl_nnz)
    tail_n = [int(l.n) for l in amg.tail_hier.levels]
    tail_nnz = [int(l.A.nnz) for l in amg.tail_hier.levels]
    sizes = [int(v) for v in lvl_n.tolist()[:amg.k]] + tail_n
    nnzs = [int(v) for v in lvl_nnz.tolist()[:amg.k]] + tail_nnz
    hier_info = {"level_sizes": sizes, "operator_complexity": round(sum(nnzs) / max(1, nnzs[0]), 3),
                 "rank_partitioned_levels": int(amg.k), "pcg": dpcg}
    # strong scaling (default): a step applies the GLOBAL preconditioner of the fixed problem once: value = steps / time.
    # weak scaling: the unit is one V-cycle over one rank's nv^3 share, a step is `world` such units
    applies_per_s = (1 if strong else world) * args.steps / elapsed
    ndof_glob = torch.tensor([st.n * bs0], dtype=torch.int64)
    dist.all_reduce(ndof_glob)
    rows_all = [None] * world
    dist.all_gather_object(rows_all, int(st.n))
    lv0 = amg.tops[0].levels[0]
    spmv_bytes = matrix_bytes(lv0.A) + 3 * 8 * lv0.A.n_rows * lv0.A.br
    k_name = "sell_spmv_kernel<EP_RES> (level 0 owned rows, rank 0)" if not elast else f"block residual kernel {lv0.A.br}x{lv0.A.br} (level 0 owned rows x [owned | ghost], rank 0)"
    k_ms = k_probe
    n_int0 = int(getattr(amg.dist_levels[0][0], "n_interior", lv0.A.n_rows))
    frac_int = n_int0 / max(1, lv0.A.n_rows)
    if k_ms is not None and k_op == 8:   # the dominant kernel of the folded cycle (same accounting as the single-GPU line), interior rows
        spmv_bytes = int(frac_int * (matrix_bytes(lv0.A) + matrix_bytes(lv0.PT) + 7 * 8 * lv0.A.n_rows + 8 * amg.tops[0].levels[1].A.n_rows))
        k_name = (f"sell_pre_restrict_kernel<512> (level 0, the {n_int0} interior rows of rank 0's {lv0.A.n_rows}: x = w Dinv b, r = b - A x, "
                  "b_c = P^T r in one pass, beside the halo exchange)")
    elif k_ms is not None and k_op == 9:
        spmv_bytes = int(frac_int * (matrix_bytes(lv0.A) + 4 * 8 * lv0.A.n_rows * lv0.A.br + (8 * lv0.A.br * lv0.A.br * lv0.A.n_rows if lv0.A.br > 1 else 0)))
        k_name = f"backward block-hybrid Gauss-Seidel sweep (level 0, interior blocks of rank 0: {n_int0} of {lv0.A.n_rows} rows, beside the halo exchange)"
    else:
        k_ms = amg.ops[0].top.time_op(0, 0, reps=50)
    achieved = spmv_bytes / (k_ms * 1e-3) / 1e9
    if rank == 0:
        out = {
            "metric": ("V-cycle applies/sec (3D H1 Poisson %s, %s V(1,1))" % ("~10M DOF" if strong and nv == 215 else f"{int(ndof_glob.item())} DOF" if strong else "~10M DOF per GPU",
                                                                                "Jacobi" if args.smoother == "jacobi" else "Gauss-Seidel")) if not elast
                      else f"V-cycle applies/sec (3D elasticity {nv}^3 nodes{'' if strong else ' per GPU'}, block size {bs0}, block-{args.smoother} V(1,1))",
            "value": round(applies_per_s, 2), "unit": "applies/s", "n_gpus": world, "rccl_ranks": int(nr.value),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "global_applies_per_s": round(args.steps / elapsed, 2),
            "config": {"workload": ((f"{args.config}: the {nv}^3-vertex grid of the single-GPU run (same matrix) cut into {world} pieces {pg}, rows per rank {rows_all}, "
                                     f"jittered Kuhn tets (seed 1), ") if strong else
                                    ((f"{args.config} per rank, weak scaling: " if elast else "weak scaling of cfg2: ") +
                                     f"global grid {tuple(pg[d] * nv for d in range(3))} = {world} x {nv}^3 vertices, hashed jitter (seed 1), ")) +
                                   (f"linear elasticity mu=1 lam=0.5, block size {bs0}, clamped left, block-{args.smoother}" if elast else f"Dirichlet right|top, {args.smoother}") +
                                   " omega=0.9, V(1,1)",
                       "parallelism": f"{world} ranks (one process per GPU), partition {pg}, {amg.k} rank-partitioned levels "
                                      f"[interior | boundary] with halo pack kernels + ncclSend/ncclRecv on a communication stream behind the C ABI "
                                      f"({ex_per_cycle:.0f} exchanges per cycle, interior rows overlap them), level {amg.k} gathered by ncclAllGather, "
                                      f"coarse hierarchy replicated from n = {amg.tail_hier.levels[0].n}; "
                                      + ("value = steps / time = applications of the GLOBAL preconditioner per second" if strong else
                                         "value = ranks x steps / time (one unit = one V-cycle over one rank's share); global_applies_per_s = steps / time"),
                       "levels": amg.k + amg.tail_hier.n_levels, "global_dof": int(ndof_glob.item()), "dist_min_rows": int(dmin),
                       "whole_cycle_graph": amg._dev.graph_info()},
            "x_norm": float(xn.item()) ** 0.5,
            "hierarchy": hier_info,
            "device_memory": {"per_rank_bytes": hier_bytes},
            "roofline": {"bound": "hbm", "kernel": k_name,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "traffic_note": "PMC counters need a profiler pass per rank; the single-GPU line carries the counter traffic of the same kernel",
                         "kernel_ms": round(k_ms, 4),
                         "kernel_ms_timing": ("HIP events around the kernel inside the running collective cycle (amgx_dist_time_kernel)" if k_in_cycle
                                              else "HIP events, back-to-back repetitions"),
                         "algorithmic_bytes": int(spmv_bytes), "cycle_algorithmic_bytes_per_rank": int(per_rank),
                         "algorithmic_model_GBs_per_gpu": round(per_rank / (ms_per_step * 1e-3) / 1e9, 1)},
        }
        cpu = None
        if not args.no_cpu_baseline:
            try:
                cpu = _dist_cpu_baseline(args, nv, world, strong, elast, hier_kw)
            except Exception as e:        # the baseline never takes the measured line down with it
                log(f"cpu_baseline of the N > 1 line failed: {e!r}")
        if cpu is not None:
            out["cpu_baseline"] = cpu
        emit(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


def _dist_cpu_baseline(args, nv, world, strong, elast, hier_kw):
    """cpu_baseline of an N > 1 line (rank 0 only, after the timed region, the other ranks wait in the final barrier): the CPU
    oracle (oracle/oracle.c: the reference's V-cycle restated in C, OpenMP over rows) on the hierarchy ONE process builds for the
    matrix the ranks share -- strong scaling: the global nv^3 matrix of the single-GPU line; weak scaling: one rank's nv^3 box,
    value scaled by 1 / world (a CPU would have to run all `world` boxes).  Bounded sample (--cpu-seconds), threads = the
    cores rank 0 may use (the other ranks are blocked in the barrier meanwhile)."""
    import __graft_entry__ as ge
    from ngsamg_amd import fem
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    if not os.environ.get("NGSAMG_NO_BUILD"):
        ge.build_oracle()
    from oracle.pyoracle import Oracle      # measured as the CPU baseline, never part of the product path
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cores = max(1, min(cores, int(int(q) // int(per))))
    except (OSError, ValueError):
        pass
    cores = max(1, min(64, cores))        # the other ranks wait in a gloo barrier (blocked on a socket, not spinning): the cores are free
    if nv ** 3 * (1 if not elast else 6 if args.config == "cfg5" else 3) > 3e7:
        log("cpu_baseline skipped: a single-process hierarchy of this size does not fit the bounded sample (see the N = 1 line)")
        return None
    t0 = time.time()
    if elast:
        rot = args.config == "cfg5"
        prob = fem.elasticity_fast((nv, nv, nv), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
        A = Matrix(prob.n, prob.n, prob.bs, prob.bs, prob.rowptr, prob.col, prob.val)
        H = Hierarchy(A, prob.free, prob.coords, dim=3, energy=1, max_coarse_size=50, regularize_cmats=0 if rot else 1, **hier_kw)
    else:
        prob = fem.poisson_fast((nv, nv, nv), dirichlet="right|top", jitter=0.2, seed=1)
        A = Matrix(prob.n, prob.n, 1, 1, prob.rowptr, prob.col, prob.val)
        H = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10, **hier_kw)
    t1 = time.time()
    n_s = prob.n * prob.bs
    b = np.random.default_rng(0).standard_normal(n_s) * np.repeat(prob.free, prob.bs)
    orc = Oracle(H.levels, sm_type="jacobi" if args.smoother == "jacobi" else "gs", omega=0.9, threads=cores)
    if cores > 1:
        orc.first_touch()
    xo = np.zeros(n_s)
    orc.apply(b, xo)
    tc0 = time.perf_counter()
    orc.apply(b, xo)
    one = time.perf_counter() - tc0
    reps = int(max(2, min(100, min(args.cpu_seconds, 15.0) / max(one, 1e-6))))
    tc0 = time.perf_counter()
    for _ in range(reps):
        orc.apply(b, xo)
    cpu_t = (time.perf_counter() - tc0) / reps
    nnz = [int(L.A.nnz) for L in H.levels]
    # the single-process hierarchy beside the rank-partitioned one (`hierarchy` of the line): sizes, OC, PCG iterations to 1e-8 --
    # what changes in the OPERATOR between the N = 1 point of a scaling curve and the N > 1 points
    sp = {"level_sizes": [int(L.n) for L in H.levels], "operator_complexity": round(sum(nnz) / max(1, nnz[0]), 3)}
    if cpu_t * 60 < 60.0:                    # ~2.5 oracle applications per PCG iteration, <= 1 minute
        try:
            sp["pcg_iterations"] = int(orc.pcg(b, tol=1e-8, maxit=200)[1])
        except Exception as e:
            log(f"oracle PCG on the single-process hierarchy failed: {e!r}")
    return {"value": round((1.0 if strong else 1.0 / world) / cpu_t, 3), "single_process_hierarchy": sp, "unit": "applies/s", "cores": int(cores), "kind": "port",
            "sample": (f"{reps} V-cycle applications of the single-process hierarchy of "
                       + (f"the same global {nv}^3 matrix" if strong else f"one rank's {nv}^3 box (value = 1 / ({world} x time): a CPU runs all {world} boxes)")
                       + f" ({prob.n * prob.bs} DOF, {H.n_levels} levels, OC {sum(nnz) / max(1, nnz[0]):.3f}; oracle/oracle.c, OpenMP over rows, "
                         f"{cores} threads on rank 0 while the other ranks wait in the final barrier; setup {t1 - t0:.1f} s, not timed)"),
            "hierarchy_note": "the rank-partitioned hierarchy (`hierarchy` above) is built by the distributed setup: same rules, rank-local aggregates"}


def _wait_ranks(procs, limit):
    """Wait for the rank processes; procs[0].stdout is read to its end.  A rank that dies, or a collective that never
    completes, must not leave the others waiting forever: on the first non-zero exit code, or after `limit` seconds, the
    remaining processes (exactly these) are killed.  Returns (exit codes, rank 0's stdout, reason or None)."""
    import threading
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    t0 = time.time()
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            why = f"a rank failed (exit codes so far {codes})"
        elif time.time() - t0 > limit:
            why = f"no result after {limit:.0f} s"
        if why:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    return codes, (buf[0] if buf else ""), why


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) BEFORE anything in this
    process touches the GPU, relay rank 0's JSON line, fail if any rank fails or fewer than N ranks joined RCCL."""
    import socket
    import subprocess
    import __graft_entry__ as ge
    if not os.environ.get("NGSAMG_NO_BUILD"):
        ge.build_host()
        ge.build_hip()
    import torch
    ndev = torch.cuda.device_count()               # (does not initialise the GPU)
    if ndev < args.gpus:
        log(f"--gpus {args.gpus} requested but only {ndev} GPU(s) are visible")
        raise SystemExit(3)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), NGSAMG_SELF_LAUNCHED="1")
        env.pop("OMP_NUM_THREADS", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    codes, out, why = _wait_ranks(procs, float(os.environ.get("NGSAMG_BENCH_TIMEOUT", "1700")))
    if why:
        log(f"multi-rank run stopped: {why}")
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(codes) or line is None:
        log(f"rank exit codes {codes}; no result" if line is None else f"rank exit codes {codes}")
        raise SystemExit(1)
    res = json.loads(line)
    if res.get("n_gpus") != args.gpus or res.get("rccl_ranks") != args.gpus:
        log(f"only {res.get('rccl_ranks')} of {args.gpus} ranks joined the RCCL communicator")
        raise SystemExit(1)
    emit(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--nv", type=int, default=None, help="vertices per direction (default 215 = cfg 2; 126 for cfg 3 / cfg 5)")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="cfg2: 3D H1 Poisson ~10M DOF (the headline metric); cfg3: 3D elasticity 126^3 nodes, 3x3 fine / 6x6 coarse blocks; "
                         "cfg5: the same with rotations, 6x6 blocks on every level (BASELINE.json configs[2], configs[4] single-GPU shape); "
                         "cfg4 (with --gpus 8): 3D H1 Poisson 342^3 = 40M DOF as 2 x 2 x 2 boxes of 171^3 vertices (BASELINE.json configs[3])")
    ap.add_argument("--smoother", default="jacobi", choices=["jacobi", "gs", "gs_mc"],
                    help="gs = Gauss-Seidel in the block-hybrid form (one launch per sweep); gs_mc = multicolour Gauss-Seidel (one launch per colour)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--edge-mats", type=int, default=0, help="cfg3 / cfg5: 1 = the energy's edge matrices + matrix-valued smoothed prolongation "
                    "(ngs_amg_edge_mats; general blocks in P instead of w Q(t)), 2 = also the energy-based strength of connection (ngs_amg_crs_robust)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the CPU baseline sample")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--ops", action="store_true", help="print HIP-event timings of the individual kernels per level")
    ap.add_argument("--scaling", default=None, choices=["strong", "weak"],
                    help="--gpus N > 1: strong (default for cfg2 / cfg4: the SAME nv^3 problem cut into N pieces, value = steps / time) or "
                         "weak (default for cfg3 / cfg5: one nv^3 box per rank, value = ranks x steps / time)")
    ap.add_argument("--dist-min-rows", type=int, default=0, help="a level stays rank-partitioned while every rank has at least this many rows")
    ap.add_argument("--hierarchy", default="spw", choices=["aaf", "spw"],
                    help="spw (measured line): the reference's and the library's default setup -- one SPW step (3 pairing rounds + orphan round) per "
                         "level, semi-aux smoothed prolongation (OC 1.54 at cfg 2); aaf: the target-driven agglomeration of rounds 1-3 (OC 1.09), a "
                         "hierarchy the reference would not build; the spw run reports it beside the measured line as `continuity`")
    ap.add_argument("--no-continuity", "--no-reference-defaults", dest="no_continuity", action="store_true",
                    help="skip the second hierarchy (the round-1..3 continuity line)")
    ap.add_argument("--multistep", action="store_true", help="cfg2, one GPU: hierarchy with ngs_amg_enable_multistep (the reference's H1 default; "
                    "not the measured configuration: denser P, fewer iterations)")
    args = ap.parse_args()

    # stdout carries exactly ONE line (the result): everything libraries print (RCCL / gloo banners) goes to stderr
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        self_launch(args)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the host setup is OpenMP-parallel: share the box's cores (or the container's CPU share, cgroup v2) between the ranks.
    # torch.distributed.run exports OMP_NUM_THREADS=1 to its children, which would make the (untimed) setup of a 10M-DOF
    # box take minutes; without any setting OpenMP would start one thread per visible core of the whole machine.
    cpus = os.cpu_count() or 8
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cpus = max(1, min(cpus, int(q) // int(per)))
    except (OSError, ValueError):
        pass
    if world > 1 or "OMP_NUM_THREADS" not in os.environ:
        os.environ["OMP_NUM_THREADS"] = os.environ.get("NGSAMG_OMP_THREADS", str(max(1, min(32, cpus // world))))
    import torch
    import __graft_entry__ as ge
    # NGSAMG_NO_BUILD=1 (tools/profile_round.sh): under rocprofv3 the profiler's preloaded library has already initialised
    # the GPU, and a compiler child (hipcc -> clang -> lld exec chain) started from here would be the exec hop the pool refuses
    if local_rank == 0 and not os.environ.get("NGSAMG_SELF_LAUNCHED") and not os.environ.get("NGSAMG_NO_BUILD"):
        ge.build_host()
        ge.build_hip()
    dist = None
    force_dist = bool(int(os.environ.get("NGSAMG_FORCE_DIST", "0")))     # world_size 1 through the distributed code path
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # torch.distributed (gloo) carries the rendezvous, the 128-byte RCCL id, the host-side setup messages and the
        # barriers of the timed region; the data path is RCCL through the C ABI (amgx_comm_create / amgx_dist_apply)
        ndev = max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank % ndev)
        dist.init_process_group(backend="gloo")
        dist.barrier()           # the libraries are (re)built by local rank 0 only
    if args.gpus != world:
        if rank == 0:
            log(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
        raise SystemExit(1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the apply path has no CPU fallback")
    device = (local_rank % max(1, torch.cuda.device_count())) if world > 1 else 0
    torch.cuda.set_device(device)

    from ngsamg_amd import fem
    from ngsamg_amd import _lib as _lib_mod
    from ngsamg_amd._lib import Matrix
    from ngsamg_amd.hierarchy import Hierarchy
    from ngsamg_amd.device import DeviceAMGMatrix, vcycle_bytes, matrix_bytes

    if args.scaling is None:
        args.scaling = "strong" if args.config in ("cfg2", "cfg4") else "weak"
    nv = args.nv if args.nv else (215 if args.config == "cfg2" else (342 if args.scaling == "strong" else 171) if args.config == "cfg4" else 126)
    if args.config == "cfg4":
        os.environ.setdefault("NGSAMG_PGRID", "box")
    if world > 1 or force_dist:
        run_distributed(args, torch, dist, world, rank, device, nv)
        return

    # ---- host setup (cold path, not timed) -------------------------------------------------------------
    # The measured line runs on the hierarchy a default-constructed ngs_amg preconditioner builds and the reference builds
    # (--hierarchy spw): ONE SPW step of 3 pairing rounds + orphan round per level (spw_agg_impl.hpp; base_factory.cpp:356-424
    # takes exactly one TryCoarseStep per level, enable_multistep is parsed but not used in that version: base_factory.cpp:27,
    # nodal_factory_impl.hpp:84) with the semi-aux smoothed prolongation (vertex_factory_impl.hpp:1836-2290) -- ~8x per level,
    # 8 levels, OC 1.54 at cfg 2.  `value`, `roofline`, `cpu_baseline` and the PCG parity block all belong to it.
    # "continuity" (below) = the hierarchy of rounds 1-3 (--hierarchy aaf: agglomerate until the level has shrunk to
    # first_aaf / aaf; OC 1.09), kept only so that the rounds stay comparable.
    hier_kw = {"spw": 1} if args.hierarchy == "spw" else {"spw": 0, "enable_multistep": int(args.multistep)}
    if args.edge_mats and args.config in ("cfg3", "cfg5"):
        hier_kw.update({"edge_mats": 1, "crs_robust": int(args.edge_mats > 1)})
    t0 = time.time()
    if args.config == "cfg2":
        prob = fem.poisson_fast((nv, nv, nv), dirichlet="right|top", jitter=0.2, seed=1)
        A = Matrix(prob.n, prob.n, 1, 1, prob.rowptr, prob.col, prob.val)
        t1 = time.time()
        H = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10, **hier_kw)
        wl = (f"cfg2: 3D P1 Poisson {nv}^3 = {prob.n} DOF, jittered Kuhn tets (seed 1), "
              f"Dirichlet right|top, {args.smoother} omega=0.9, V(1,1), max_coarse_size=50")
    else:
        rot = args.config == "cfg5"
        prob = fem.elasticity_fast((nv, nv, nv), dirichlet="left", mu=1.0, lam=0.5, rotations=rot)
        A = Matrix(prob.n, prob.n, prob.bs, prob.bs, prob.rowptr, prob.col, prob.val)
        t1 = time.time()
        H = Hierarchy(A, prob.free, prob.coords, dim=3, energy=1, max_coarse_size=50, regularize_cmats=0 if rot else 1, **hier_kw)
        wl = (f"{args.config}: 3D linear elasticity {nv}^3 nodes = {prob.n * prob.bs} DOF, mu=1 lam=0.5, "
              f"{'displacements + rotations, 6x6 blocks on every level' if rot else '3x3 blocks on level 0, 6x6 below'}, clamped left, "
              f"block-{args.smoother} omega=0.9, V(1,1), max_coarse_size=50")
    bs0 = prob.bs
    free_s = np.repeat(prob.free, bs0).astype(np.float64)
    n_s = prob.n * bs0
    t2 = time.time()
    dev_sm = {"jacobi": "jacobi", "gs": "hgs", "gs_mc": "gs"}[args.smoother]
    torch.cuda.synchronize()
    mem0 = torch.cuda.mem_get_info(device)[0]
    amg = DeviceAMGMatrix(H, sm_type=dev_sm, omega=0.9, mg_cycle="V", clev="inv", device=device,
                          use_graph=not args.no_graph)
    torch.cuda.synchronize()
    hier_bytes = int(mem0 - torch.cuda.mem_get_info(device)[0])       # device memory the uploaded hierarchy holds (all copies)
    t3 = time.time()
    if rank == 0:
        log(f"assembly {t1 - t0:.1f}s, hierarchy {t2 - t1:.1f}s, upload {t3 - t2:.1f}s")
        log(H.summary().replace("\n", "\n[bench] "))
    cycle_bytes, per_level = vcycle_bytes(H)

    # ---- timed region: K applications, inputs resident in HBM ------------------------------------------
    rng = np.random.default_rng(0)
    b_host = rng.standard_normal(n_s) * free_s
    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        b = torch.from_numpy(b_host).to(f"cuda:{device}")
        x = torch.empty_like(b)
        for _ in range(args.warmup):
            amg.Mult(b, x)
        stream.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ts = time.perf_counter()
        for _ in range(args.steps):
            amg.Mult(b, x)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        te = time.perf_counter()
    elapsed = te - ts
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{device}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    x_norm = float(torch.linalg.norm(x).item())
    applies_per_s = world * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel ---------------------------------------------------------------
    # Jacobi cycle: sell_pre_restrict_kernel on level 0 = Jacobi pre-smoothing from zero, residual and restriction in one
    # pass.  Algorithmic bytes per launch (the un-fused op sequence it replaces):
    #   x = w Dinv b: 3 V_0;  r = b - A x: B(A_0) + 3 V_0;  b_c = P^T r: B(PT_0) + V_0 + V_1
    # Gauss-Seidel cycle (or no fused kernel): level-0 residual SpMV r = b - A x: B(A_0) + 3 V_0
    lv0 = H.levels[0]
    V0 = 8 * lv0.n * lv0.bs
    with torch.cuda.stream(stream):
        amg.Mult(b, x)
        stream.synchronize()
    folded = amg.matrix_info(0, "Q")["fmt"] is not None
    k_name, tname = "sell_spmv_kernel<EP_RES> (level 0: r = b - A x)", "traffic_spmv_l0.json"
    spmv_bytes = matrix_bytes(lv0.A) + 3 * V0
    k_ms = k_ms_b2b = None
    if args.smoother == "jacobi" and args.config == "cfg2":
        try:
            # timed INSIDE the cycle (HIP events around the one kernel while whole cycles run): the figure rocprofv3's
            # kernel trace reports for it; the back-to-back repetition time is kept beside it
            k_ms = amg.time_op(0, 8, reps=50)
            if args.ops:          # back-to-back repetitions only on request: they would mix into the rocprofv3 --stats average
                k_ms_b2b = amg.time_op(0, 7, reps=50)
            spmv_bytes = matrix_bytes(lv0.A) + matrix_bytes(lv0.PT) + 7 * V0 + 8 * H.levels[1].n * H.levels[1].bs
            k_name = "sell_pre_restrict_kernel<512> (level 0: x = w Dinv b, r = b - A x, b_c = P^T r in one pass)"
            tname = "traffic_pre_restrict_l0.json"
        except Exception as e:
            log(f"in-cycle timing of the fused kernel unavailable: {e!r}")
            k_ms = None
    in_cycle = k_ms is not None
    if args.smoother == "gs" and args.config == "cfg2" and lv0.bs == 1:
        # Gauss-Seidel cycle: the backward block-hybrid sweep of level 0 (x' = GS_back(x + P x_c)) is the largest kernel.
        # Algorithmic bytes of one sweep in the RHS form ("each read A once"): B(A_0) + dinv + x in, b, x out
        try:
            k_ms = amg.time_op(0, 9, reps=50)
            spmv_bytes = matrix_bytes(lv0.A) + 4 * V0
            k_name = "gsb_sweep_kernel<256, 1, false> (level 0: backward block-hybrid Gauss-Seidel sweep, x' = GS(x + P x_c))"
            tname = "traffic_gsb_sweep_l0.json"
            in_cycle = True
        except Exception as e:
            log(f"in-cycle timing of the Gauss-Seidel sweep unavailable: {e!r}")
            k_ms = None
    if args.smoother == "gs" and args.config != "cfg2" and lv0.bs > 1:
        # block levels: the backward block-hybrid sweep (bgsb_sweep_kernel) timed inside the cycle, same byte model
        try:
            k_ms = amg.time_op(0, 9, reps=30)
            spmv_bytes = matrix_bytes(lv0.A) + 4 * V0 + 8 * lv0.bs * lv0.bs * lv0.n        # + the inverted block diagonals
            k_name = f"bgsb_sweep_kernel<{bs0}, false> (level 0: backward block-hybrid Gauss-Seidel sweep, {bs0}x{bs0} blocks)"
            tname = f"traffic_bgsb_sweep_{args.config}.json"
            in_cycle = True
        except Exception as e:
            log(f"in-cycle timing of the block Gauss-Seidel sweep unavailable: {e!r}")
            k_ms = None
    if k_ms is None:
        k_ms = amg.time_op(0, 0, reps=50)
        if args.config != "cfg2":
            k_name = f"bsell_spmv_kernel<{bs0}, EP_RES> (level 0: r = b - A x, {bs0}x{bs0} blocks)"
            tname = f"traffic_bsell_res_{args.config}.json"
    achieved = spmv_bytes / (k_ms * 1e-3) / 1e9
    traffic = traffic_src = None
    if args.hierarchy == "spw":          # (the counters were collected per hierarchy: the level-0 kernels stream different P^T / Q)
        tname = tname.replace(".json", "_spw.json")
    tpath = os.path.join(ROOT, "profiles", tname)
    if os.path.exists(tpath) and nv == (215 if args.config == "cfg2" else 126):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch")
            # PMC counters cannot be collected from inside this process: the figure comes from a separate rocprofv3 --pmc
            # pass (profiles/), valid for the kernel build named here
            traffic_src = {"file": "profiles/" + tname, "commit": tj.get("commit"), "collected": tj.get("collected")}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": k_name,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel_ms": round(k_ms, 4), "kernel_ms_timing": "HIP events around the kernel inside the running cycle (amgx_time_op op 8 / 9)" if in_cycle else "HIP events, back-to-back repetitions",
                "kernel_ms_back_to_back": round(k_ms_b2b, 4) if k_ms_b2b else None,
                "algorithmic_bytes": int(spmv_bytes)}
    # whole cycle: bytes the cycle actually streams (device encodings) / time.  The algorithmic byte count of the reference's
    # op sequence divided by the same time is NOT a bandwidth for the folded cycle (it skips a pass over A and the
    # pass over P): it is reported as the speed-up measure it is.
    if folded and args.config == "cfg2":
        sb = 0
        for l in range(H.n_levels - 1):
            Vl, Vc = 8 * H.levels[l].n * H.levels[l].bs, 8 * H.levels[l + 1].n * H.levels[l + 1].bs
            # (the images the cycle actually reads: the local-window forms of A' and Q where a level has them)
            def _sb(plain, lw):
                i = amg.matrix_info(l, lw)
                return i["stream_bytes"] if i["fmt"] is not None else amg.matrix_info(l, plain)["stream_bytes"]
            sb += _sb("Apre", "ApreLW") + amg.matrix_info(l, "PT")["stream_bytes"] + _sb("Q", "QLW") + 5 * Vl + 2 * Vc
    elif args.config != "cfg2":
        # block levels: folded where Q exists (A once + PT + Q), literal otherwise (A twice + P + PT), device encodings
        sb = 0
        for l in range(H.n_levels - 1):
            Vl, Vc = 8 * H.levels[l].n * H.levels[l].bs, 8 * H.levels[l + 1].n * H.levels[l + 1].bs
            mi = {w: amg.matrix_info(l, w) for w in ("A", "P", "PT", "Q")}
            if mi["Q"]["fmt"] is not None:
                sb += mi["A"]["stream_bytes"] + mi["PT"]["stream_bytes"] + mi["Q"]["stream_bytes"] + 8 * Vl + 2 * Vc
            else:
                sb += 2 * mi["A"]["stream_bytes"] + mi["P"]["stream_bytes"] + mi["PT"]["stream_bytes"] + 15 * Vl + 2 * Vc
    else:
        sb = cycle_bytes
    roofline["cycle_streamed_bytes"] = int(sb)
    roofline["cycle_GBs"] = round(sb / (ms_per_step * 1e-3) / 1e9, 1)
    roofline["cycle_frac"] = round(sb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    roofline["algorithmic_model"] = {"cycle_bytes": int(cycle_bytes), "bytes_over_time_GBs": round(cycle_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                                     "speedup_vs_literal_bytes": round(cycle_bytes / sb, 3),
                                     "note": "bytes of the reference's un-fused op sequence over the measured time: a speed-up measure, not a bandwidth"}

    if args.ops and rank == 0:
        names = {0: "residual r=b-Ax", 1: "jacobi fused", 2: "restrict PT r", 3: "prolong x+P xc"}
        for l in range(H.n_levels - 1):
            qi = amg.matrix_info(l, "Q")
            log(f"level {l} cycle step down (pre-smooth + restrict) {amg.time_op(l, 5, reps=50) * 1e3:9.1f} us   "
                f"up (correct + post-smooth) {amg.time_op(l, 6, reps=50) * 1e3:9.1f} us   Q fmt={qi['fmt']} lanes={qi['lanes']} "
                f"stream={qi['stream_bytes'] / 1e6:.1f} MB")
        for l in range(H.n_levels):
            for op in range(4):
                if op >= 2 and l + 1 >= H.n_levels:
                    continue
                if l + 1 >= H.n_levels and op == 1:
                    continue
                try:
                    ms = amg.time_op(l, op, reps=50)
                except Exception:            # (e.g. the fused Jacobi step on a Gauss-Seidel level)
                    continue
                M = H.levels[l].A if op < 2 else (H.levels[l].PT if op == 2 else H.levels[l].P)
                by = matrix_bytes(M) + 8 * H.levels[l].n * H.levels[l].bs * (3 if op == 0 else 4 if op == 1 else 1 if op == 2 else 2)
                info = amg.matrix_info(l, "A" if op < 2 else ("PT" if op == 2 else "P"))
                log(f"level {l} {names[op]:18s} {ms * 1e3:9.1f} us  {by / ms / 1e6:8.1f} GB/s  fmt={info['fmt']} lanes={info['lanes']} stream={info['stream_bytes'] / 1e6:.1f} MB")
        log(f"whole cycle (graph) {amg.time_op(0, 4, reps=50) * 1e3:.1f} us")

    # ---- CPU baseline: the oracle (restatement of the reference's cycle) on this box's cores ----------
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        if not os.environ.get("NGSAMG_NO_BUILD"):
            ge.build_oracle()
        from oracle.pyoracle import Oracle      # measured as the CPU baseline, never part of the product path
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        cores = min(cores, 64)                   # more threads than memory channels only adds OpenMP overhead
        try:                                     # a container's CPU share (cgroup v2 cpu.max) is the real core count:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]      # threads beyond it are only throttled
            if q != "max":
                cores = max(1, min(cores, int(int(q) // int(per))))
        except (OSError, ValueError):
            pass
        # Gauss-Seidel: the CPU baseline is the reference's sequential sweep; parity of the GPU result is checked against
        # the oracle run in the GPU's own order (block-hybrid: tests/hgs_oracle.py, multicolour: gs_mc)
        orc = Oracle(H.levels, sm_type="jacobi" if args.smoother == "jacobi" else "gs", omega=0.9, threads=cores)
        if cores > 1:
            orc.first_touch()                    # NUMA placement: every thread first-writes the rows it streams
        xo = np.zeros(n_s)
        orc.apply(b_host, xo)                    # warm-up (first touch of the work vectors)
        tc0 = time.perf_counter()
        orc.apply(b_host, xo)                    # duration estimate
        one = time.perf_counter() - tc0
        reps = int(max(2, min(100, args.cpu_seconds / max(one, 1e-6))))
        tc0 = time.perf_counter()
        for _ in range(reps):
            orc.apply(b_host, xo)
        cpu_t = (time.perf_counter() - tc0) / reps
        if args.smoother == "gs":
            from tests.hgs_oracle import hgs_levels
            plv, ptypes = hgs_levels(H.levels, amg.hgs)
            xp = Oracle(plv, sm_type=ptypes, omega=0.9, threads=cores).apply(b_host)
        elif args.smoother == "gs_mc":
            xp = Oracle(H.levels, sm_type="gs_mc", omega=0.9, threads=cores).apply(b_host)
        else:
            xp = xo
        parity = float(np.linalg.norm(x.cpu().numpy() - xp) / np.linalg.norm(xp))
        # north_star: "iteration count and residual norm" of the Krylov solve the preconditioner sits in, GPU and CPU:
        # PCG to 1e-8 on A x = b, both with the criterion of the reference's drivers (err_k = sqrt(<C r_k, r_k>))
        pcg = None
        try:
            from ngsamg_amd.krylov import CGSolver
            with torch.cuda.stream(stream):
                cg = CGSolver(amg, amg, tol=1e-8, maxsteps=200)
                xs = cg.Solve(b)
                rt = torch.empty_like(b)
                amg.MatVec(0, xs, rt)
                fm = torch.from_numpy(free_s).to(b.device)     # the system lives on the free dofs
                g_res = float((torch.linalg.norm((b - rt) * fm) / torch.linalg.norm(b)).item())
                stream.synchronize()
            xc, c_it, c_errs = orc.pcg(b_host, tol=1e-8, maxit=200)
            A0 = H.levels[0].A.to_scipy()
            c_res = float(np.linalg.norm((b_host - A0 @ xc) * free_s) / np.linalg.norm(b_host))
            pcg = {"tol": 1e-8, "converges": bool(cg.iterations < 200 and g_res < 1e-5), "gpu_iterations": int(cg.iterations), "cpu_iterations": int(c_it),
                   "gpu_rel_residual": g_res, "cpu_rel_residual": c_res,
                   "solution_rel_diff": float(np.linalg.norm(xs.cpu().numpy() - xc) / np.linalg.norm(xc))}
            try:
                # time to solution of the whole solve with everything on the device (amgx_pcg: SpMV, cycle and BLAS-1 kernels)
                from ngsamg_amd.krylov import NativeCGSolver
                with torch.cuda.stream(stream):
                    ncg = NativeCGSolver(amg, amg, tol=1e-8, maxsteps=200)
                    ncg.Solve(b)                               # warm-up (graph capture of the cycle for these vectors)
                    stream.synchronize()
                    t_s = time.perf_counter()
                    ncg.Solve(b)
                    stream.synchronize()
                    pcg["native_solve_ms"] = round(1e3 * (time.perf_counter() - t_s), 3)
                    pcg["native_iterations"] = int(ncg.iterations)
                    # the same solve in the single-reduction form of the recurrence (AMGX_PCG_SINGLE_REDUCTION)
                    scg = NativeCGSolver(amg, amg, tol=1e-8, maxsteps=200, single_reduction=True)
                    scg.Solve(b)
                    stream.synchronize()
                    t_s = time.perf_counter()
                    scg.Solve(b)
                    stream.synchronize()
                    pcg["native_single_reduction_ms"] = round(1e3 * (time.perf_counter() - t_s), 3)
                    pcg["native_single_reduction_iterations"] = int(scg.iterations)
                    # the same system by restarted GMRES(30) (amgx_gmres; criterion |C r_k| <= tol |C r_0|)
                    from ngsamg_amd.krylov import NativeGMResSolver
                    ngm = NativeGMResSolver(amg, amg, tol=1e-8, maxsteps=200, restart=30)
                    ngm.Solve(b)
                    stream.synchronize()
                    t_s = time.perf_counter()
                    xg = ngm.Solve(b)
                    stream.synchronize()
                    pcg["native_gmres_ms"] = round(1e3 * (time.perf_counter() - t_s), 3)
                    pcg["native_gmres_iterations"] = int(ngm.iterations)
                    amg.MatVec(0, xg, rt)
                    pcg["native_gmres_rel_residual"] = float((torch.linalg.norm((b - rt) * fm) / torch.linalg.norm(b)).item())
            except Exception as e:
                log(f"native PCG timing failed: {e!r}")
            log(f"PCG to 1e-8: GPU {cg.iterations} iterations (|b - A x| / |b| on the free dofs = {g_res:.2e}), CPU oracle {c_it} ({c_res:.2e})")
        except Exception as e:                   # the parity block must never cost the bench line
            log(f"PCG parity block failed: {e!r}")
        # 1-thread leg = what ONE MPI rank of the reference does on its share
        one_t = None
        try:
            Oracle.set_threads(1)
            t1a = time.perf_counter()
            orc.apply(b_host, xo)
            d1 = time.perf_counter() - t1a
            r1 = int(max(1, min(5, 6.0 / max(d1, 1e-6))))
            t1a = time.perf_counter()
            for _ in range(r1):
                orc.apply(b_host, xo)
            one_t = {"value": round(r1 / (time.perf_counter() - t1a), 3), "cores": 1, "sample": f"{r1} applications, same oracle object limited to one thread"}
            Oracle.set_threads(cores)
        except Exception as e:
            log(f"1-thread baseline failed: {e!r}")
        cpu = {"value": round(1.0 / cpu_t, 3), "unit": "applies/s", "cores": cores, "kind": "port", "one_thread": one_t,
               "sample": f"{reps} V-cycle applications of the same {n_s}-DOF hierarchy (oracle/oracle.c, OpenMP over rows, first-touch placement)",
               "GBs_algorithmic": round(cycle_bytes / cpu_t / 1e9, 1), "gpu_vs_oracle_rel_err": parity, "pcg": pcg}
        log(f"cpu baseline: {1.0 / cpu_t:.2f} applies/s on {cores} threads; GPU-vs-oracle rel. error {parity:.2e}")

    # ---- continuity: the hierarchy of rounds 1-3 beside the measured (default = reference) one ---------------------------
    cont = None
    if rank == 0 and args.hierarchy == "spw" and not args.no_continuity and not args.multistep:
        try:
            tr0 = time.time()
            if args.config == "cfg2":
                H2 = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10, spw=0)
            else:
                H2 = Hierarchy(A, prob.free, prob.coords, dim=3, energy=1, max_coarse_size=50, regularize_cmats=0 if args.config == "cfg5" else 1, spw=0)
            amg2 = DeviceAMGMatrix(H2, sm_type=dev_sm, omega=0.9, mg_cycle="V", clev="inv", device=device, use_graph=not args.no_graph)
            tr1 = time.time()
            with torch.cuda.stream(stream):
                x2 = torch.empty_like(b)
                for _ in range(args.warmup):
                    amg2.Mult(b, x2)
                stream.synchronize()
                tq = time.perf_counter()
                for _ in range(args.steps):
                    amg2.Mult(b, x2)
                stream.synchronize()
                el2 = time.perf_counter() - tq
                from ngsamg_amd.krylov import CGSolver as _CG
                cg2 = _CG(amg2, amg2, tol=1e-8, maxsteps=200)
                cg2.Solve(b)
                cg1 = _CG(amg, amg, tol=1e-8, maxsteps=200)
                cg1.Solve(b)
                stream.synchronize()
            cb2, _ = vcycle_bytes(H2)
            cont = {"hierarchy": "aaf-driven agglomeration of rounds 1-3 (spw = 0): a hierarchy the reference would not build; round-to-round continuity only",
                    "value": round(args.steps / el2, 2), "unit": "applies/s", "ms_per_step": round(1e3 * el2 / args.steps, 4),
                    "levels": H2.n_levels, "level_sizes": [int(l.n) for l in H2.levels], "OC": round(H2.operator_complexity(), 3),
                    "pcg_iterations": int(cg2.iterations), "pcg_iterations_measured_line": int(cg1.iterations),
                    "pcg_time_model_ms": {"continuity": round(cg2.iterations * 1e3 * el2 / args.steps, 2),
                                          "measured_line": round(cg1.iterations * ms_per_step, 2),
                                          "note": "iterations x cycle time (the cycle's share of a PCG solve to 1e-8)"},
                    "algorithmic_cycle_bytes": int(cb2), "setup_s": round(tr1 - tr0, 1)}
            log(f"continuity (aaf hierarchy): {cont['value']} applies/s, OC {cont['OC']}, levels {cont['level_sizes']}, PCG {cg2.iterations} vs {cg1.iterations} iterations")
            del amg2, x2
        except Exception as e:
            log(f"continuity block failed: {e!r}")

    # what an unmodified host-pointer caller gets (vectors cross PCIe in both directions inside the call): never `value`
    host_rate = None
    try:
        xh = np.empty(n_s)
        amg.Mult(b_host, xh)
        th = time.perf_counter()
        for _ in range(5):
            amg.Mult(b_host, xh)
        host_rate = round(5.0 / (time.perf_counter() - th), 1)
    except Exception as e:
        log(f"host-pointer probe failed: {e!r}")

    # Gauss-Seidel: iteration counts of PCG (1e-8) with the GPU's sweep order and with the reference's sequential order
    gs_its = None
    if args.smoother in ("gs", "gs_mc") and rank == 0 and not args.no_cpu_baseline:
        try:
            from oracle.pyoracle import Oracle as _O
            if args.smoother == "gs":
                it_gpu = _O(plv, sm_type=ptypes, threads=cores).pcg(b_host, tol=1e-8, maxit=200)[1]
            else:
                it_gpu = _O(H.levels, sm_type="gs_mc", threads=cores).pcg(b_host, tol=1e-8, maxit=200)[1]
            it_seq = _O(H.levels, sm_type="gs", threads=cores).pcg(b_host, tol=1e-8, maxit=200)[1]
            gs_its = {"tol": 1e-8, "gpu_order_iterations": int(it_gpu), "sequential_order_iterations": int(it_seq)}
        except Exception as e:
            log(f"GS iteration comparison failed: {e!r}")

    if rank == 0:
        out = {
            "metric": ("V-cycle applies/sec (3D H1 Poisson ~10M DOF, %s V(1,1))" % ("Jacobi" if args.smoother == "jacobi" else "Gauss-Seidel")) if args.config == "cfg2"
                      else f"V-cycle applies/sec (3D elasticity {nv}^3 nodes, block size {bs0}, block-{args.smoother} V(1,1))",
            "value": round(applies_per_s, 2), "unit": "applies/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            # (one GPU: the N = 1 point of the strong-scaling series bench.py --gpus N runs for cfg 2 / cfg 4, of the weak one for cfg 3 / cfg 5)
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl + (" [enable_multistep]" if args.multistep else "")
                                   + ((" [edge_mats: matrix-valued smoothed prolongation" + (", crs_robust" if args.edge_mats > 1 else "") + "]") if hier_kw.get("edge_mats") else ""),
                       "level_sizes": [int(l.n) for l in H.levels],
                       "levels": H.n_levels, "operator_complexity": round(H.operator_complexity(), 3),
                       "nnz_level0": lv0.A.nnz, "graph_replay": not args.no_graph,
                       "post_smoothing": ("folded into the prolongation: x' = z + (I - w Dinv A) P x_c, same result up to rounding "
                                          "(AMGX_NO_FOLD=1 runs the literal kernel sequence)") if folded else "literal",
                       "parallelism": "1 GPU"},
            # cold path, untimed: host assembly / hierarchy setup (Galerkin products on the device when NGSAMG_DEVICE_SETUP != 0) / amgx_create
            "setup_s": {"assembly": round(t1 - t0, 2), "hierarchy": round(t2 - t1, 2), "amgx_create": round(t3 - t2, 2),
                        "galerkin_on_device": bool(_lib_mod._device_setup)},
            "x_norm": x_norm,
            "device_memory": {"hierarchy_bytes": hier_bytes, "level0_csr_bytes": int(matrix_bytes(lv0.A)),
                              "ratio_to_level0_csr": round(hier_bytes / max(1, matrix_bytes(lv0.A)), 2)},
            "host_ptr_applies_per_s": host_rate,
            "roofline": roofline,
        }
        if gs_its is not None:
            out["gs_iterations"] = gs_its
        if cont is not None:
            out["continuity"] = cont
        out["config"]["hierarchy"] = ("aaf-driven agglomeration of rounds 1-3 (spw = 0): NOT the reference's hierarchy, continuity only"
                                      if args.hierarchy == "aaf" else
                                      "library default = the reference's setup rules: one SPW step (3 pairing rounds + orphan round) per level, semi-aux smoothed prolongation")
        if cpu is not None:
            out["cpu_baseline"] = cpu
            if cpu.get("pcg") is not None:
                # (block-Jacobi with omega = 0.9 is not a convergent smoother on the rotational model problem of cfg 5: the
                #  applications/s of such a line is the throughput of a cycle, not of a solver)
                out["converges"] = cpu["pcg"].get("converges")
        emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
