"""Hybrid-LSQR 512^2 x 180 with x_true given (bench.py's C3 form: every iterate formed), fixed lambda: eight timed solves."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
R = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
x = torch.rand(N * N, device="cuda"); b = R.apply(x)
b = b + 0.01 * torch.randn_like(b) * b.norm() / b.numel() ** 0.5
Hybrid_LSQR(R, b, 5, 1e-2, x_true=x, history=False)
ts = []
for _ in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    Hybrid_LSQR(R, b, 100, 1e-2, x_true=x, history=False)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("fixed lambda, x_true:", " ".join(f"{100/t:.0f}" for t in ts), "it/s")
