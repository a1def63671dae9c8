import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
import torch
from ngsamg_amd import fem, Matrix
from ngsamg_amd.hierarchy import Hierarchy
from ngsamg_amd.device import DeviceAMGMatrix
p = fem.poisson_fast((215,215,215), dirichlet="right|top", jitter=0.2, seed=1)
A = Matrix(p.n, p.n, 1, 1, p.rowptr, p.col, p.val)
H = Hierarchy(A, p.free, p.coords, dim=3, energy=0, max_coarse_size=50, max_levels=10)
amg = DeviceAMGMatrix(H, sm_type="jacobi", device=0)
rng = np.random.default_rng(0)
b = rng.standard_normal(p.n) * p.free
x = np.empty(p.n)
for _ in range(3): amg.Mult(b, x)
t=time.perf_counter()
for _ in range(20): amg.Mult(b, x)
dt=(time.perf_counter()-t)/20
print(f"host vectors (pageable numpy, H2D + cycle + D2H + sync): {dt*1e3:.2f} ms per application = {1/dt:.1f} applies/s")
bp = torch.from_numpy(b).pin_memory(); xp = torch.empty(p.n, dtype=torch.float64).pin_memory()
bn, xn = bp.numpy(), xp.numpy()
for _ in range(3): amg.Mult(bn, xn)
t=time.perf_counter()
for _ in range(20): amg.Mult(bn, xn)
dt=(time.perf_counter()-t)/20
print(f"host vectors (pinned): {dt*1e3:.2f} ms per application = {1/dt:.1f} applies/s")
bd = torch.from_numpy(b).cuda(); xd = torch.empty_like(bd)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): amg.Mult(bd, xd)
    s.synchronize(); t=time.perf_counter()
    for _ in range(100): amg.Mult(bd, xd)
    s.synchronize(); dt=(time.perf_counter()-t)/100
print(f"device vectors, graph replay: {dt*1e3:.3f} ms")
amg2 = DeviceAMGMatrix(H, sm_type="jacobi", device=0, use_graph=False)
with torch.cuda.stream(s):
    for _ in range(5): amg2.Mult(bd, xd)
    s.synchronize(); t=time.perf_counter()
    for _ in range(100): amg2.Mult(bd, xd)
    s.synchronize(); dt=(time.perf_counter()-t)/100
print(f"device vectors, direct launches (no graph): {dt*1e3:.3f} ms")
