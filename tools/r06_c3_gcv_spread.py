"""C3 (512^2 x 180) Hybrid-LSQR with gcv / dp: the time of every one of 12 consecutive 100-iteration solves (outliers?)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Radon2DParallel
from trips_py_amd.solvers import Hybrid_LSQR
N = 512
A = Radon2DParallel(N, np.linspace(0, np.pi, 180, endpoint=False))
dev = A.engine.device
g = torch.Generator(device=dev).manual_seed(1)
xt = torch.rand(N * N, device=dev, generator=g)
b = A.apply(xt)
e = torch.randn(b.numel(), device=dev, generator=g)
delta = 0.01 * float(b.norm())
b = b + e * (delta / e.norm())
for reg, kw in (("gcv", {}), ("dp", {"delta": delta})):
    Hybrid_LSQR(A, b, 100, reg, xt, **kw)
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        Hybrid_LSQR(A, b, 100, reg, xt, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(reg, " ".join(f"{100 / t:.0f}" for t in ts), "it/s")
