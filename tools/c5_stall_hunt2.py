"""The sporadic ~80 ms stall of C5 solves: does it need operators being created / destroyed around the solve?"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import BlockDiagOp, Radon2DParallel
from trips_py_amd.solvers import CGLS
mode = sys.argv[1] if len(sys.argv) > 1 else "destroy"
Nf, nt, na = 256, 32, 15
keep = []
def one(i):
    ops = [Radon2DParallel(Nf, np.deg2rad(t + 12.0 * np.arange(na))) for t in range(nt)]
    F = BlockDiagOp(ops)
    if mode == "keep":
        keep.append((ops, F))
    x = torch.rand(F.shape[1], device="cuda"); b = F.apply(x); x0 = torch.zeros_like(x)
    CGLS(F, b, x0, 5, 0, history=False)
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        CGLS(F, b, x0, 100, 0, history=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out
for i in range(10):
    t0 = time.perf_counter()
    r = one(i)
    print(mode, i, " ".join(f"{v:7.2f}" for v in r), f"ms; whole call {(time.perf_counter() - t0) * 1e3:7.1f} ms; gc {gc.get_count()}")
