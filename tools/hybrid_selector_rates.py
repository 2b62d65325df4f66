"""Hybrid-LSQR with the automatic selectors: does limiting the host BLAS pool matter, and how much does the time jitter?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); b = R.apply(x)
b = b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5
delta = float(0.01 * b.norm())
for reg, kw in (("gcv", {}), ("dp", {"delta": delta})):
    Hybrid_LSQR(R, b, 5, reg, x_true=x, history=False, **kw)
    ts = []
    for _ in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Hybrid_LSQR(R, b, 100, reg, x_true=x, history=False, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(reg, " ".join(f"{100/t:.0f}" for t in ts), "it/s")
