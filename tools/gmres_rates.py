"""Hybrid-GMRES on the 512^2 blur (a9): iterations/s with fixed lambda, GCV and the discrepancy principle, 60-iteration solves."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import Hybrid_GMRES
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
x = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
b = A.apply(x)
e = torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
delta = 0.01 * float(b.norm())
b = b + e * (delta / e.norm())
for reg, kw in ((1e-2, {}), ("gcv", {}), ("dp", {"delta": delta})):
    Hybrid_GMRES(A, b, 5, reg, x, **kw)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xx, info = Hybrid_GMRES(A, b, 60, reg, x, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(reg, " ".join(f"{60 / t:.0f}" for t in ts), "it/s  relError[-1] %.4f" % info["relError"][-1])
