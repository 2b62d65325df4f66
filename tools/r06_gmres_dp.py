"""Hybrid-GMRES with the discrepancy principle on the 512^2 blur through the one-call loop: iterations/s and the library's phase timers by
the number of worker threads; the Python loop for comparison."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from trips_py_amd.operators import Blur2D
from trips_py_amd.problems import gauss_psf
from trips_py_amd.solvers import Hybrid_GMRES
N = 512
A = Blur2D(gauss_psf((9, 9), (3, 3))[0], N, N)
dev = A.engine.device
x = torch.rand(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
b = A.apply(x)
e = torch.randn(N * N, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
delta = 0.01 * float(b.norm())
b = b + e * (delta / e.norm())
for name, kw in [("python loop", {"c_loop": False})] + [(f"one call per iteration, {w} worker(s)", {"search_workers": w}) for w in (1, 2, 3, 4)]:
    Hybrid_GMRES(A, b, 20, "dp", x, delta=delta, **kw)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        xx, info = Hybrid_GMRES(A, b, 60, "dp", x, delta=delta, **kw)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ph = []
    if "c_loop" not in kw:
        Hybrid_GMRES(A, b, 60, "dp", x, delta=delta, host_phases=ph, **kw)
        ph = [f"{1e6 * v / 60:.1f}" for v in ph]
    print(f"{name:40s}", " ".join(f"{60 / t:6.0f}" for t in ts), "it/s", ("  us per iteration: wait step | enqueue | wait worker | post | launch x: " + " ".join(ph)) if ph else "",
          " lambda[-1] %.6e" % info["regParam"])
