#!/usr/bin/env python3
"""How much of a long-row level's SpMV time is the gather of x?  The level-1 matrix of the cfg-2 default hierarchy (1.24 M rows x 52)
with its real columns, with perfectly coalesced columns (col = row - len/2 + k) and with random columns; same values, same
row lengths, same image format.   python tools/gather_probe.py [nv]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ngsamg_amd import fem
from ngsamg_amd._lib import Matrix
from ngsamg_amd.hierarchy import Hierarchy
from ngsamg_amd.device import DeviceAMGMatrix

nv = int(sys.argv[1]) if len(sys.argv) > 1 else 215
prob = fem.poisson_fast((nv, nv, nv), dirichlet="right|top", jitter=0.2, seed=1)
A = Matrix(prob.n, prob.n, 1, 1, prob.rowptr, prob.col, prob.val)
H = Hierarchy(A, prob.free, prob.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10)
L1 = H.levels[1]
rp = np.asarray(L1.A.rowptr)
n = L1.n
ln = np.diff(rp)
real = np.array(L1.A.col, copy=True)
rows = np.repeat(np.arange(n, dtype=np.int64), ln)
k = np.arange(rp[-1], dtype=np.int64) - np.repeat(rp[:-1], ln)
start = np.clip(np.arange(n, dtype=np.int64) - ln // 2, 0, n - ln)
coal = (np.repeat(start, ln) + k).astype(np.int32)
rng = np.random.default_rng(0)
def run(tag, cols):
    L1.A.col[:] = cols
    amg = DeviceAMGMatrix(H, sm_type="jacobi", omega=0.9, mg_cycle="V", clev="inv", device=0)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        t0 = amg.time_op(1, 0, reps=50)
        t5 = amg.time_op(1, 5, reps=50)
        t6 = amg.time_op(1, 6, reps=50)
    info = amg.matrix_info(1, "A")
    print(f"{tag:28s} residual {t0*1e3:7.1f} us   down {t5*1e3:7.1f} us  up {t6*1e3:7.1f} us   fmt={info['fmt']} lanes={info['lanes']} stream={info['stream_bytes']/1e6:.1f} MB", flush=True)
    del amg
run("real columns", real)
run("coalesced columns", coal)
# random: per row a sorted random sample containing the diagonal
r = rng.integers(0, n, size=rp[-1]).astype(np.int64)
r[rp[:-1] + ln // 2] = np.arange(n)
order = np.lexsort((r, rows))
r = r[order]
# duplicates inside a row are harmless for timing but check_matrix wants ascending columns: nudge duplicates
d = np.diff(r, prepend=-1)
same = (d == 0) & (np.diff(rows, prepend=-1) == 0)
r[same] = np.minimum(r[same] + 1, n - 1)
run("random columns", r.astype(np.int32))
