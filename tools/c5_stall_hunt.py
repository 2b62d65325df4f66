"""Hunt for the sporadic ~80 ms stall in C5 solves: 40 CGLS solves, host-return time and device-done time of each."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
from trips_py_amd.solvers import CGLS
Nf, nt, na = 256, 32, 15
ops = [Radon2DParallel(Nf, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)]
F = BlockDiagOp(ops)
x = torch.rand(F.shape[1], device="cuda"); b = F.apply(x); x0 = torch.zeros_like(x)
CGLS(F, b, x0, 5, 0, history=False)
torch.cuda.synchronize()
rows = []
for i in range(40):
    t0 = time.perf_counter()
    CGLS(F, b, x0, 100, 0, history=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    rows.append((t1 - t0, t2 - t0))
for i, (a, c) in enumerate(rows):
    flag = "  <-- stall" if c > 0.012 else ""
    if flag or i < 3:
        print(f"solve {i:2d}: CGLS() returned after {a*1e3:7.2f} ms, device idle after {c*1e3:7.2f} ms{flag}")
print("median", sorted(c for _, c in rows)[20] * 1e3, "ms; stalls:", sum(c > 0.012 for _, c in rows))
